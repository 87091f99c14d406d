/* Hook points of oracle/hooked/thread_task.patch.  TEST INFRASTRUCTURE ONLY.
 *
 * oracle/_ref_hooked/libdav1d_hooked.so is the reference build with ONE file changed: a copy of src/thread_task.c, made at build time
 * and patched (oracle/hooked/thread_task.patch) at exactly the places INTEGRATION.md 2 names — the tile task's call of
 * dav1d_decode_tile_sbrow (reference src/thread_task.c:733-752), the publication of a frame's rows (:888-896) and the two places a
 * frame is declared complete (:780-790, :899-913).  Everything else — dav1d_submit_frame, dav1d_decode_frame_init, the task
 * queues, check_tile's inter-frame dependencies, dav1d_worker_task on its own threads — is the reference's code as it lies under
 * /root/reference.  The hooks are what oracle/ref_hooked.c plugs in: the injection of a synthetic pass-1 output (there are no AV1
 * streams in either box) and, in HIP mode, the calls of include/dav1d_hip.h that INTEGRATION.md tells a maintainer to make. */
#ifndef DAV1D_ORACLE_HOOKS_H
#define DAV1D_ORACLE_HOOKS_H
#include "src/internal.h"

typedef struct Dav1dHooks {
    /* after dav1d_decode_frame_init + dav1d_decode_frame_init_cdf, on the worker that ran them, before the frame's tile tasks exist */
    int (*after_init)(Dav1dFrameContext *f);
    /* instead of dav1d_decode_tile_sbrow with pass 1 (entropy decoding): the hand-off arrays were injected by after_init */
    int (*entropy_tile_sbrow)(Dav1dTaskContext *t);
    /* instead of dav1d_decode_tile_sbrow with pass 2; NULL: the reference's own pass 2 */
    int (*recon_tile_sbrow)(Dav1dTaskContext *t);
    /* the frame's last task is through.  Non-NULL: the backend finishes the frame on a thread of its own and calls
     * dav1d_hooked_frame_done (which publishes the rows and runs dav1d_decode_frame_exit); rows are then NOT published per
     * superblock row by the task loop, since the pixels do not exist before that */
    void (*frame_complete)(Dav1dFrameContext *f);
} Dav1dHooks;

extern const Dav1dHooks *dav1d_hooks;                 /* NULL: the unpatched behaviour */
void dav1d_hooked_frame_done(Dav1dFrameContext *f, int retval);
/* the first `rows` luma rows of the frame's picture are final (src/thread_task.c:888-896 for a backend that finishes frames itself) */
void dav1d_hooked_rows_done(Dav1dFrameContext *f, unsigned rows);
#endif
