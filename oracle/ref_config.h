/* Hand-written build configuration for compiling the reference's C path
 * (asm disabled) as the parity oracle.  It restates the symbols the reference's
 * meson.build would generate (meson.build:62-87,171-175,198,226-271,338,499-501)
 * for x86-64 Linux, HAVE_ASM=0, both bit depths, no DSP trimming.
 * TEST INFRASTRUCTURE ONLY: nothing in the product library includes this. */
#ifndef ORACLE_REF_CONFIG_H
#define ORACLE_REF_CONFIG_H
#define ARCH_AARCH64 0
#define ARCH_ARM 0
#define ARCH_LOONGARCH 0
#define ARCH_LOONGARCH64 0
#define ARCH_PPC64LE 0
#define ARCH_RISCV 0
#define ARCH_RV32 0
#define ARCH_RV64 0
#define ARCH_X86 1
#define ARCH_X86_32 0
#define ARCH_X86_64 1
#define CONFIG_16BPC 1
#define CONFIG_8BPC 1
#define CONFIG_LOG 1
#define ENDIANNESS_BIG 0
#define HAVE_ASM 0
#define HAVE_AS_FUNC 0
#define HAVE_ALIGNED_ALLOC 1
#define HAVE_C11_GENERIC 1
#define HAVE_CLOCK_GETTIME 1
#define HAVE_DLSYM 1
#define HAVE_ELF_AUX_INFO 0
#define HAVE_GETAUXVAL 0
#define HAVE_MEMALIGN 1
#define HAVE_POSIX_MEMALIGN 1
#define HAVE_PTHREAD_GETAFFINITY_NP 1
#define HAVE_PTHREAD_NP_H 0
#define HAVE_PTHREAD_SETAFFINITY_NP 1
#define HAVE_PTHREAD_SETNAME_NP 1
#define HAVE_PTHREAD_SET_NAME_NP 0
#define HAVE_SIGACTION 1
#define HAVE_SYS_TYPES_H 1
#define HAVE_UNISTD_H 1
#define TRIM_DSP_FUNCTIONS 0
#endif
