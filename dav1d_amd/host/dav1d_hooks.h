/* Hook points of patches/dav1d-1.5.4-hip.patch (goes to dav1d's src/ with it).
 *
 * The patch changes ONE file of dav1d, src/thread_task.c, at exactly the places INTEGRATION.md 2 names — the start of a frame's init task (:700-702), the tile task's call of
 * dav1d_decode_tile_sbrow (reference src/thread_task.c:733-752), the publication of a frame's rows (:888-896) and the places a frame is
 * declared complete (:780-790, :876-885, :899-913) — and adds the two functions a backend that finishes frames on its own thread calls
 * back.  Everything else — dav1d_submit_frame, dav1d_decode_frame_init, the task queues, check_tile's inter-frame dependencies,
 * dav1d_worker_task on its own threads — is dav1d's code.  dav1d_hooks == NULL: the unpatched behaviour. */
#ifndef DAV1D_HIP_HOOKS_H
#define DAV1D_HIP_HOOKS_H
#include "src/internal.h"

typedef struct Dav1dHooks {
    /* before dav1d_decode_frame_init, on the worker that is about to run it: the frame context still holds what its last frame left — the moment for
     * a backend to let go of that frame's objects if the frame never reached frame_complete (it failed in pass 1) and work of its own on the
     * context's arrays (f->frame_thread.cf ...) may still be running: dav1d_decode_frame_init is free to reallocate them.  NULL: nothing */
    void (*before_init)(Dav1dFrameContext *f);
    /* after dav1d_decode_frame_init + dav1d_decode_frame_init_cdf, on the worker that ran them, before the frame's tile tasks exist */
    int (*after_init)(Dav1dFrameContext *f);
    /* instead of dav1d_decode_tile_sbrow with pass 1 (entropy decoding); NULL: dav1d's own */
    int (*entropy_tile_sbrow)(Dav1dTaskContext *t);
    /* instead of dav1d_decode_tile_sbrow with pass 2; NULL: dav1d's own pass 2 */
    int (*recon_tile_sbrow)(Dav1dTaskContext *t);
    /* the frame's last task is through WITHOUT an error.  Non-NULL: the backend finishes the frame on a thread of its own and calls
     * dav1d_hip_frame_done (which publishes the rows and runs dav1d_decode_frame_exit); rows are then NOT published per superblock row
     * by the task loop, since the pixels do not exist before that.  A frame with task_thread.error set never gets here: the task loop
     * ends it itself (dav1d_decode_frame_exit with the error, FRAME_ERROR in progress[1]) as it always did. */
    void (*frame_complete)(Dav1dFrameContext *f);
} Dav1dHooks;

extern const Dav1dHooks *dav1d_hooks;                 /* NULL: the unpatched behaviour */
/* the frame the backend took over has ended: retval 0 or a negative errno (DAV1D_ERR(EINVAL): its bitstream or a reference was bad) */
void dav1d_hip_frame_done(Dav1dFrameContext *f, int retval);
/* the first `rows` luma rows of the frame's picture are final (src/thread_task.c:888-896 for a backend that finishes frames itself) */
void dav1d_hip_rows_done(Dav1dFrameContext *f, unsigned rows);
#endif
