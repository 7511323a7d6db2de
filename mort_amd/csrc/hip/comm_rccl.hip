/*
 * comm_rccl.hip -- the one exchange step of the multi-GPU render: every rank renders its interleaved row
 * blocks (mort_hip_set_partition), then the packed uchar4 rows are gathered to rank 0 over RCCL (xGMI inside a node) and
 * de-interleaved into the full framebuffer on rank 0's GPU.  The reference is single-GPU (device 0 only, textures.cuh:91);
 * this is the north_star's "RCCL only for the final framebuffer gather".
 *
 * Shape of the exchange: xGMI is point to point (each peer has its own link to rank 0), so the gather is one group of
 * ncclRecv on rank 0 / one ncclSend per peer -- every link carries exactly its rank's rows once (W x H x 4 / N bytes:
 * 0.4 MB per peer for 1200x675 on 8 GPUs), no ring.  One process per GPU: the C CLI forks its ranks before any HIP call
 * (mort.c), Python uses torch.distributed for the same step (mort_amd/partition.py).
 *
 * librccl is loaded on first use (dlopen), so single-GPU users never map it.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "mort_ctx.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    if (!r.handle) {
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
            r.Send = (decltype(r.Send))dlsym(r.handle, "ncclSend");
            r.Recv = (decltype(r.Recv))dlsym(r.handle, "ncclRecv");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
        }
    }
    return r;
}
int rccl_fail(mort_ctx *c, ncclResult_t e, const char *what) {
    if (c) c->last_error = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
    return MORT_ERR_HIP;
}
#define RCHK(ctx, call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return rccl_fail(ctx, e_, #call); } while (0)

int rows_of(int rank, int nranks, int rpb, int height) {
    int n = 0;
    const int nblocks = (height + rpb - 1) / rpb;
    for (int b = rank; b < nblocks; b += nranks) { int r0 = b * rpb, r1 = r0 + rpb; if (r1 > height) r1 = height; n += r1 - r0; }
    return n;
}

} // namespace

/* gathered[r][local row][x] -> frame[global row][x]: one thread per pixel of the frame */
__global__ void __launch_bounds__(256) deinterleave_kernel(const uchar4 *gathered, uchar4 *frame, int width, int height, int nranks, int rpb, size_t tile_stride_px) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)width * (size_t)height) return;
    const int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * (size_t)width);
    const int block = y / rpb, within = y - block * rpb;
    const int r = block % nranks, lb = block / nranks;
    const int ly = lb * rpb + within;
    frame[i] = gathered[(size_t)r * tile_stride_px + (size_t)ly * (size_t)width + (size_t)x];
}

extern "C" int mort_hip_comm_id(void *id) {
    if (!id) return MORT_ERR_INVALID;
    if (!rccl().ok) return MORT_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return MORT_ERR_HIP;
    std::memcpy(id, &u, MORT_COMM_ID_BYTES);
    return MORT_OK;
}

extern "C" int mort_hip_comm_init(mort_ctx *c, const void *id, int rank, int nranks) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return MORT_ERR_INVALID;
    if (c->part.rank != rank || c->part.nranks != nranks) return MORT_ERR_INVALID; /* set the row partition first */
    if (!rccl().ok) { c->last_error = "librccl.so could not be loaded"; return MORT_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) { rccl().CommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    ncclUniqueId u;
    std::memcpy(&u, id, MORT_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCHK(c, rccl().CommInitRank(&comm, nranks, u, rank));
    c->comm = (void *)comm;
    return MORT_OK;
}

extern "C" void mort_hip_comm_destroy(mort_ctx *c) {
    if (c && c->comm && rccl().ok) { hipSetDevice(c->device); rccl().CommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c) { hipFree(c->d_gather); hipFree(c->d_frame); c->d_gather = c->d_frame = nullptr; c->gather_cap = c->frame_cap = 0; }
}

/* render + gather as ONE stream of work: render kernel(s) -> send / recv group -> de-interleave -> copy out, all enqueued on the
 * context's stream, one host wait at the end (the render's statistics are collected after it).  Failure discipline (the
 * exchange is collective): everything that can fail locally -- argument checks, allocations -- happens BEFORE the render; once the
 * send / recv group is entered it is always closed (GroupEnd), and a rank whose render failed still takes part in the exchange
 * (its rows are whatever its buffer holds) and reports its error afterwards, so that no peer is left waiting in ncclRecv. */
extern "C" int mort_hip_render_gather(mort_ctx *c, const mort_camera *cam, int mode, uint8_t *rgba_out, mort_stats *stats) {
    if (!c || !cam) return MORT_ERR_INVALID;
    const int W = cam->image_width, H = cam->image_height;
    if (W <= 0 || H <= 0) return MORT_ERR_INVALID;
    const int N = c->part.nranks, R = c->part.rank, rpb = c->part.rows_per_block;
    if (N > 1 && !c->comm) return MORT_ERR_INVALID;
    if (R == 0 && !rgba_out) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const int lr = rows_of(R, N, rpb, H);
    const size_t max_rows = (size_t)(((H + rpb - 1) / rpb + N - 1) / N) * (size_t)rpb; /* every rank's tile fits */
    const size_t tile_px = max_rows * (size_t)W;
    /* ---- every allocation first ---- */
    if (c->rgba_cap < tile_px * 4) {
        if (c->d_rgba) { hipFree(c->d_rgba); c->d_rgba = nullptr; c->rgba_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_rgba, tile_px * 4));
        c->rgba_cap = tile_px * 4;
    }
    if (N > 1 && R == 0) {
        if (c->gather_cap < tile_px * 4 * (size_t)N) {
            hipFree(c->d_gather); c->d_gather = nullptr; c->gather_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_gather, tile_px * 4 * (size_t)N));
            c->gather_cap = tile_px * 4 * (size_t)N;
        }
        if (c->frame_cap < (size_t)W * H * 4) {
            hipFree(c->d_frame); c->d_frame = nullptr; c->frame_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_frame, (size_t)W * H * 4));
            c->frame_cap = (size_t)W * H * 4;
        }
    }
    if (!c->ev_g0) HIPCHK(c, hipEventCreate(&c->ev_g0));
    if (!c->ev_g1) HIPCHK(c, hipEventCreate(&c->ev_g1));
    /* ---- render: enqueued, not waited for ---- */
    mort_stats local;
    std::memset(&local, 0, sizeof local);
    c->defer_stats = true; c->pending_stats = nullptr;
    const int st_render = mort_hip_render_device(c, cam, mode, c->d_rgba, nullptr, c->stream, &local);
    c->defer_stats = false;
    /* ---- the exchange: entered by every rank whatever its render returned ---- */
    int st_x = MORT_OK;
    ncclResult_t e_x = ncclSuccess;
    hipError_t h_x = hipEventRecord(c->ev_g0, c->stream);
    if (N == 1) {
        if (st_render == MORT_OK && h_x == hipSuccess) h_x = hipMemcpyAsync(rgba_out, c->d_rgba, (size_t)W * H * 4, hipMemcpyDeviceToHost, c->stream);
    } else {
        if (R == 0 && h_x == hipSuccess) h_x = hipMemcpyAsync(c->d_gather, c->d_rgba, (size_t)lr * W * 4, hipMemcpyDeviceToDevice, c->stream);
        e_x = rccl().GroupStart();
        if (e_x == ncclSuccess) {
            if (R == 0) {
                for (int r = 1; r < N && e_x == ncclSuccess; r++) {
                    const size_t bytes = (size_t)rows_of(r, N, rpb, H) * (size_t)W * 4;
                    if (bytes) e_x = rccl().Recv((unsigned char *)c->d_gather + (size_t)r * tile_px * 4, bytes, ncclUint8, r, (ncclComm_t)c->comm, c->stream);
                }
            } else if (lr > 0) {
                e_x = rccl().Send(c->d_rgba, (size_t)lr * (size_t)W * 4, ncclUint8, 0, (ncclComm_t)c->comm, c->stream);
            }
            const ncclResult_t e_end = rccl().GroupEnd(); /* always closed: a failed Send / Recv must not leave the group open for the next call */
            if (e_x == ncclSuccess) e_x = e_end;
        }
        if (R == 0 && e_x == ncclSuccess && h_x == hipSuccess) {
            const size_t npx = (size_t)W * (size_t)H;
            hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, (const uchar4 *)c->d_gather, (uchar4 *)c->d_frame,
                               W, H, N, rpb, tile_px);
            h_x = hipGetLastError();
            if (h_x == hipSuccess) h_x = hipMemcpyAsync(rgba_out, c->d_frame, npx * 4, hipMemcpyDeviceToHost, c->stream);
        }
    }
    if (h_x == hipSuccess) h_x = hipEventRecord(c->ev_g1, c->stream);
    const hipError_t h_sync = hipStreamSynchronize(c->stream); /* the one host wait */
    if (e_x != ncclSuccess) st_x = rccl_fail(c, e_x, "frame gather (ncclSend / ncclRecv group)");
    else if (h_x != hipSuccess) st_x = hip_fail(c, h_x, "frame gather");
    else if (h_sync != hipSuccess) st_x = hip_fail(c, h_sync, "hipStreamSynchronize(frame gather)");
    if (st_render != MORT_OK) { c->pending_stats = nullptr; return st_render; }
    if (st_x != MORT_OK) { c->pending_stats = nullptr; return st_x; }
    if (c->pending_stats) {
        const int st_s = c->pending_stats(&local);
        c->pending_stats = nullptr;
        if (st_s != MORT_OK) return st_s;
    }
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_g0, c->ev_g1));
    local.gather_seconds = ms * 1e-3;
    if (stats) *stats = local;
    return MORT_OK;
}

/* One-rank rehearsal of everything the gather uses (librccl loaded, symbols bound, communicator, grouped ncclSend / ncclRecv of
 * uchar rows on the context's stream, de-interleave kernel): rank 0 sends a tile to itself.  A one-GPU box cannot host two
 * RCCL ranks, so this is what the GPU tests can run of the RCCL path; the N-rank exchange itself runs on a multi-GPU node. */
extern "C" int mort_hip_comm_selftest(mort_ctx *c) {
    if (!c) return MORT_ERR_INVALID;
    if (!rccl().ok) { c->last_error = "librccl.so could not be loaded"; return MORT_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    RCHK(c, rccl().GetUniqueId(&u));
    ncclComm_t comm = nullptr;
    RCHK(c, rccl().CommInitRank(&comm, 1, u, 0));
    const int W = 256, H = 64;
    const size_t bytes = (size_t)W * H * 4;
    unsigned char *d_a = nullptr, *d_b = nullptr, *d_f = nullptr;
    int rc = MORT_OK;
    std::vector<unsigned char> h(bytes), back(bytes);
    for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 2654435761u >> 13);
    if (hipMalloc((void **)&d_a, bytes) != hipSuccess || hipMalloc((void **)&d_b, bytes) != hipSuccess || hipMalloc((void **)&d_f, bytes) != hipSuccess) rc = MORT_ERR_NOMEM;
    if (rc == MORT_OK && hipMemcpyAsync(d_a, h.data(), bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MORT_ERR_HIP;
    if (rc == MORT_OK) {
        ncclResult_t e = rccl().GroupStart();
        if (e == ncclSuccess) e = rccl().Send(d_a, bytes, ncclUint8, 0, comm, c->stream);
        if (e == ncclSuccess) e = rccl().Recv(d_b, bytes, ncclUint8, 0, comm, c->stream);
        if (e == ncclSuccess) e = rccl().GroupEnd(); else rccl().GroupEnd();
        if (e != ncclSuccess) rc = rccl_fail(c, e, "self send/recv");
    }
    if (rc == MORT_OK) { /* one rank: the de-interleave is the identity */
        hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)((size_t)W * H + 255) / 256), dim3(256), 0, c->stream, (const uchar4 *)d_b, (uchar4 *)d_f, W, H, 1, 8, (size_t)W * H);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(back.data(), d_f, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) rc = MORT_ERR_HIP;
        else if (std::memcmp(back.data(), h.data(), bytes) != 0) { c->last_error = "self send/recv returned different bytes"; rc = MORT_ERR_HIP; }
    }
    hipFree(d_a); hipFree(d_b); hipFree(d_f);
    rccl().CommDestroy(comm);
    return rc;
}
