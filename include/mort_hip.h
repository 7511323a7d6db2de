/*
 * mort_hip.h -- C ABI of libmort_hip.so, the MI355X (gfx950) render path.
 *
 * The reference has no plugin / FFI interface: host and device share one
 * translation unit and the boundary is a kernel launch with by-value structs.
 * The entry points below are what a binding for that boundary has to replace:
 *
 *   mort_hip_upload_world   <- world::toDevice()            world.cuh:98-102
 *                              (16x cudaMemcpyToSymbol:     objects.cuh:848-856,
 *                               materials.cuh:264-270, textures.cuh:320-325,
 *                               image upload textures.cuh:89-127)
 *   mort_hip_rng_seed       <- setup_rng<<<>>>              rng.cuh:8-15, mort.cu:706-709
 *   mort_hip_rng_load/store <- (dev_states array itself)    mort.cu:706-708
 *   mort_hip_render[_device]<- renderKernel<<<>>> + the four per-bounce scratch
 *                              allocations                  mort.cu:44-47,106,712-725
 *   mort_hip_init/shutdown  <- implicit device 0 context    textures.cuh:91
 *
 * Conventions: plain C, no C++ types or exceptions cross the boundary.  The
 * caller owns every host buffer; the library owns all device memory behind
 * the opaque mort_ctx (except buffers passed to mort_hip_render_device).
 * Calls are blocking unless stated; one context per process per device; a
 * context is not re-entrant.  Every function returns MORT_OK (0) or a
 * negative mort_status; the reference's convention is print-and-exit
 * (HANDLE_ERROR, include/book.h:21-30), which the `mort` CLI reproduces on
 * top of these codes.
 */
#ifndef MORT_HIP_H
#define MORT_HIP_H

#include "mort_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mort_ctx mort_ctx;

typedef enum mort_status {
    MORT_OK = 0,
    MORT_ERR_INVALID = -1,      /* NULL / out-of-range argument */
    MORT_ERR_NO_DEVICE = -2,    /* no HIP device, or not gfx950 */
    MORT_ERR_HIP = -3,          /* a HIP runtime call failed (see mort_hip_last_error) */
    MORT_ERR_NO_WORLD = -4,     /* render before upload_world */
    MORT_ERR_NO_RNG = -5,       /* render before rng_seed / rng_load */
    MORT_ERR_UNSUPPORTED = -6,  /* scene graph outside what the kernels flatten */
    MORT_ERR_CAPACITY = -7,     /* bounce_limit > MORT_MAX_BOUNCE_LIMIT, nesting too deep, ... */
    MORT_ERR_NOMEM = -8
} mort_status;

/* render modes */
#define MORT_MODE_MEGA 0 /* one lane per pixel, whole path on chip */
#define MORT_MODE_THROUGHPUT 2 /* NOT a drop-in, NOT bit-compatible with the reference: the reference keys one cuRAND stream per pixel (mort.cu:23-24), which
                                 * makes a pixel's samples one serial chain; this mode keys one stream per (pixel, stratum row) -- subsequence x + (y*sqrt_spp + s_j)*W of
                                 * the same seed -- so a pixel's sqrt_spp rows run as independent work items, and a resolve kernel sums them in order.  Same estimator,
                                 * same per-sample arithmetic, deterministic and partition-invariant, but different random numbers: images agree with MORT_MODE_MEGA only
                                 * statistically.  Needs mort_hip_rng_seed (the seed is reused; rng_load'ed states are not); worlds the two LDS state-machine
                                 * kernels take (scenes 1, 10; the final scenes 8, 9), else MORT_ERR_UNSUPPORTED; the per-pixel states of the other modes are left untouched.  bench.py reports it only
                                 * as a separately labelled line */
#define MORT_MODE_WAVE 1 /* wavefront (queued) pipeline: every world the megakernels stage in LDS (all ten built-in scenes), else MORT_ERR_UNSUPPORTED; blocking */

/* Which rows of the image this context renders: rows are grouped into blocks
 * of `rows_per_block` and this context owns blocks rank, rank+nranks, ...
 * (image-space partition across GPUs).  {0,1,8} = whole image. */
typedef struct mort_partition {
    int rank;
    int nranks;
    int rows_per_block; /* multiple of 8 */
} mort_partition;

typedef struct mort_stats {
    double seconds;          /* device time of the render kernel(s), HIP events */
    uint64_t segments;       /* world::hit calls == path segments traced */
    uint64_t pixels;         /* pixels rendered by this context */
    uint64_t eff_samples;    /* pixels * floor(sqrt(spp))^2 */
    uint64_t rng_draws;      /* curand_uniform calls */
    uint64_t algorithmic_hbm_bytes; /* 100 B/pixel (+12 with accum) */
    int scene_in_lds;        /* 1 if the kernel staged the scene in LDS */
    int local_rows;          /* rows owned under the partition */
    int kernel_vgprs, kernel_lds_bytes; /* launch facts, for reports */
    uint64_t reference_walks; /* BVH megakernel: segments re-traced with the reference's own walk */
    char kernel_name[64];    /* dominant kernel of this render as rocprofv3 names it (template arguments included) */
    double gather_seconds;   /* mort_hip_render_gather: device time of the frame gather (RCCL + de-interleave + copy out) */
} mort_stats;

const char *mort_hip_strerror(int status);
/* Text of the last HIP error seen by this context ("" if none). */
const char *mort_hip_last_error(const mort_ctx *ctx);

/* device = HIP ordinal (one process per GPU uses LOCAL_RANK). */
int mort_hip_init(int device, mort_ctx **out);
void mort_hip_shutdown(mort_ctx *ctx);

/* Copies and re-lays-out the world for the kernels; the caller keeps ownership
 * of every host array (including image texels) and may free them afterwards. */
int mort_hip_upload_world(mort_ctx *ctx, const mort_world *world);

int mort_hip_set_partition(mort_ctx *ctx, const mort_partition *part);

/* states[x + y*W] = curand_init(seed, subsequence = x + y*W, offset = 0) for
 * every pixel this context owns (unlike rng.cuh:8-15, bounds-guarded). */
int mort_hip_rng_seed(mort_ctx *ctx, uint64_t seed, int width, int height);
/* Full-image arrays of W*H 48-byte curandStateXORWOW records, row-major from
 * the bottom row (the reference's layout); only owned rows are read/written. */
int mort_hip_rng_load(mort_ctx *ctx, const mort_rng_state *states, int width, int height);
int mort_hip_rng_store(mort_ctx *ctx, mort_rng_state *states, int width, int height);

/* Renders the owned rows into full-size host buffers: rgba_out W*H*4 bytes
 * (uchar4, row 0 = bottom row, as the reference's GL buffer), accum_out W*H*3
 * floats (linear pixel mean before gamma) or NULL, segments_px_out W*H
 * uint32 or NULL.  Rows not owned are left untouched. */
int mort_hip_render(mort_ctx *ctx, const mort_camera *cam, int mode, uint8_t *rgba_out, float *accum_out,
                    uint32_t *segments_px_out, mort_stats *stats);

/* Same render into caller-provided DEVICE buffers holding only the owned rows,
 * packed (local_rows * W pixels): d_rgba local_rows*W*4 bytes, d_accum
 * local_rows*W*3 floats or NULL.  Launched on `stream` (a hipStream_t, NULL =
 * the context's stream); asynchronous when stats is NULL, otherwise waits and
 * fills stats.  This is the entry bench.py and the multi-GPU gather use. */
int mort_hip_render_device(mort_ctx *ctx, const mort_camera *cam, int mode, void *d_rgba, void *d_accum,
                           void *stream, mort_stats *stats);

/* ---- `mort --mode host` (BASELINE config 1; the CPU figure timed beside the GPU): the render path as a straight host
 * loop of the SAME kernel body the GPU runs (camera.cuh:86-242 and everything below it), on `nthreads` host threads.
 * An explicit mode, never a fallback; needs no GPU and makes no HIP runtime call.  `states`: W*H records, read and
 * written in place (full image, row 0 = bottom row).  flags: MORT_HOST_TREE = walk this build's unified tree where the
 * world has one (what one GPU lane does) instead of the reference's scan over every primitive. ---- */
#define MORT_HOST_TREE 1
int mort_hip_rng_seed_host(uint64_t seed, int width, int height, mort_rng_state *states);
int mort_hip_render_host(const mort_world *world, const mort_camera *cam, mort_rng_state *states, int nthreads, int flags,
                         uint8_t *rgba_out, float *accum_out, uint32_t *segments_px_out, mort_stats *stats);

/* ---- multi-GPU: one process (one context) per GPU, image rows partitioned with mort_hip_set_partition, ONE exchange step:
 * the packed uchar4 rows of every rank are gathered to rank 0 over RCCL and de-interleaved there (the reference is
 * single-GPU).  Rank 0 obtains an id (mort_hip_comm_id) and hands its MORT_COMM_ID_BYTES bytes to the other ranks by any
 * means (the CLI uses pipes to the ranks it forked); every rank then calls mort_hip_comm_init (collective) after
 * mort_hip_set_partition(rank, nranks, rows_per_block).  mort_hip_render_gather = render + gather, collective, blocking;
 * rank 0 receives the full W*H*4 frame in rgba_out (other ranks may pass NULL).  With nranks == 1 no communicator is needed. */
#define MORT_COMM_ID_BYTES 128
int mort_hip_comm_id(void *id);
int mort_hip_comm_init(mort_ctx *ctx, const void *id, int rank, int nranks);
void mort_hip_comm_destroy(mort_ctx *ctx);
int mort_hip_render_gather(mort_ctx *ctx, const mort_camera *cam, int mode, uint8_t *rgba_out, mort_stats *stats);
/* One-rank rehearsal of the RCCL path (library load, communicator, grouped send / recv to self, de-interleave): MORT_OK when the bytes come back. */
int mort_hip_comm_selftest(mort_ctx *ctx);

/* ---- roofline calibration (measurement only; bench.py prints the results beside the render kernels' counters).
 * mort_hip_calib_valu: shader cycles one SIMD needs per wave64 VALU instruction with exactly `waves_per_simd` (1..8) waves
 * resident on every SIMD; kind 0 = independent v_fma_f32, 1 = one dependent v_fma_f32 chain, 2 = independent v_fma_f64,
 * 3 = three v_fma_f32 per scalar instruction, 4 = independent v_pk_fma_f32 (two fma per lane and instruction).  mort_hip_calib_hbm_copy: GB/s (read + write) of a float4 copy of `bytes`
 * per buffer (use > 256 MB, the Infinity Cache), best of `reps`. ---- */
typedef struct mort_calib_valu {
    int waves_per_simd, kind;
    double seconds;                  /* HIP-event time of the launch */
    double cycles_per_wave;          /* s_memtime ticks around the loop, median over waves */
    double clock_ghz;                /* s_memtime ticks per s_memrealtime tick (100 MHz), median over waves */
    double valu_per_wave;            /* VALU instructions each wave issued */
    int simds_seen;                  /* distinct (XCC, SE, SH, CU, SIMD) the waves reported (HW_REG_HW_ID / HW_REG_XCC_ID) */
    double resident_waves_per_simd;  /* waves of this launch that really overlapped on a SIMD (median over SIMDs) */
    double cycles_per_valu_per_wave; /* what ONE wave sustains: cycles_per_wave / valu_per_wave */
    double cycles_per_valu_per_simd; /* what a SIMD sustains: (first start .. last end of its waves) x clock / instructions issued on it, median over SIMDs */
} mort_calib_valu;
int mort_hip_calib_valu(mort_ctx *ctx, int waves_per_simd, int kind, mort_calib_valu *out);
int mort_hip_calib_hbm_copy(mort_ctx *ctx, size_t bytes, int reps, double *gbs_out);

/* Number of rows owned for an image of `height` rows under the current partition. */
int mort_hip_local_rows(const mort_ctx *ctx, int height);
/* Global row index of local row `local_row`. */
int mort_hip_global_row(const mort_ctx *ctx, int local_row);

#ifdef __cplusplus
}
#endif
#endif /* MORT_HIP_H */
