// csr_build.hip - the two segment lists of a batch (CSR over end hits and over start hits), built on the GPU.
//
// The reference keeps the hit <-> segment association as dense one-hot matrices Ri, Ro [N, E] built per graph on the
// host (gnn/graph.py:28-35) and padded per batch (gnn/trainSegmentClassifier.py:66-95); its on-disk form is
// `Ri.nonzero()` / `Ro.nonzero()` in row-major order (gnn/graph.py:20-26): for every hit the ids of the segments that
// end (start) there, ascending.  gnn_csr_build makes exactly those arrays from the index form (src, dst): the same
// arrays hitgraph._csr_by builds with a stable sort on the host, entry for entry - what the per-module kernels
// (k_node, the backward walks) consume.  A never-seen batch used to pay two stable torch sorts and a read-back here
// (0.45 ms for one 100k-segment graph: most of a trigger-style "one graph in, scores out" call,
// gnn/Inference.ipynb cell 3); this is four or six small launches and no read-back.
//
//   k_csr_degrees   one lane per segment: atomic counts per end hit / start hit; malformed segments set the status word
//   k_csr_scan*     exclusive scan of both count arrays -> in_ptr / out_ptr  (one workgroup per array up to 256k hits,
//                   else block sums -> their scan -> block rescan)
//   k_csr_fill      one lane per segment: claims a slot of its end hit's / start hit's list (atomic countdown on the
//                   counts) - the order INSIDE a list depends on the atomics' arrival
//                   (batches of >= 1 M segments: k_csr_degrees_wg / k_csr_fill_wg - a workgroup takes 16 k
//                   consecutive segments and, when their endpoints fall into a range of < 16 k hits (segments
//                   stored graph by graph do), counts and claims in LDS: one global atomic per touched hit instead
//                   of one per segment - device-scope atomics run at ~26 G/s on this part, 2 ms for c3 x 256)
//   k_csr_rank      one lane per segment: its rank in its list = number of ids in the list smaller than its own
//                   (lists are short: ~10 entries) -> the final slot; ascending ids whatever the arrival order was,
//                   so the arrays (and every sum the kernels make over them) are the same in every run
#include "common.h"

#include <cstdlib>

namespace gnn {
namespace {

constexpr int kScanItems = 8;                          // per thread of a scan block
constexpr int kScanTile = kBlock * kScanItems;         // 2048 counts per workgroup
constexpr int64_t kOneBlockMax = 262144;               // up to here one workgroup scans an array in a loop
constexpr int64_t kWgMinSegments = 1 << 20;            // from here on: LDS-private counting (k_csr_*_wg)

__device__ __forceinline__ bool seg_ok(int s, int d, int64_t n) { return (unsigned)s < (unsigned)n && (unsigned)d < (unsigned)n; }

__global__ __launch_bounds__(kBlock) void k_csr_degrees(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                        int64_t n_hits, int64_t n_segments, int32_t *__restrict__ deg_in,
                                                        int32_t *__restrict__ deg_out, int32_t *__restrict__ status)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_segments) return;
    const int s = src[j], d = dst[j];
    if (seg_ok(s, d, n_hits)) {
        atomicAdd(deg_in + d, 1);
        atomicAdd(deg_out + s, 1);
    } else if (!(s < 0 && d < 0)) {
        atomicOr(status, 1);                           // an end outside [0, n_hits), or exactly one end negative
    }
}

// inclusive scan of one value per thread over the workgroup (NT threads); returns the thread's inclusive sum,
// *total = the workgroup's sum
template <int NT = kBlock>
__device__ __forceinline__ int block_scan_incl(int v, int *total)
{
    __shared__ int wsum[NT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    __syncthreads();                                   // (wsum of a previous call has been read)
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
        if (i < w) before += wsum[i];
        all += wsum[i];
    }
    *total = all;
    return v + before;
}

// one workgroup per array (blockIdx.x: 0 = in, 1 = out): ptr[0 .. n] = exclusive scan of deg[0 .. n)
// (no __restrict__: the block sums of a large array are scanned in place)
// (1024 threads: a 10 k-hit graph is two passes of the loop, not five - this kernel is latency, not throughput)
constexpr int kScanOneThreads = 1024, kScanOneTile = kScanOneThreads * kScanItems;
__global__ __launch_bounds__(kScanOneThreads) void k_csr_scan_one(const int32_t *deg2, int32_t *in_ptr, int32_t *out_ptr,
                                                                  int64_t n, int64_t stride)
{
    const int32_t *deg = deg2 + blockIdx.x * stride;
    int32_t *ptr = blockIdx.x ? out_ptr : in_ptr;
    int carry = 0;
    for (int64_t base = 0; base < n; base += kScanOneTile) {
        const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
        int v[kScanItems], sum = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            v[k] = i0 + k < n ? deg[i0 + k] : 0;
            sum += v[k];
        }
        int total;
        int run = carry + block_scan_incl<kScanOneThreads>(sum, &total) - sum;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            if (i0 + k < n) ptr[i0 + k] = run;
            run += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) ptr[n] = carry;
}

// large arrays, step 1: sums of the 2048-count blocks (grid: blocks x 2)
__global__ __launch_bounds__(kBlock) void k_csr_scan_sums(const int32_t *__restrict__ deg2, int64_t n, int64_t stride,
                                                          int32_t *__restrict__ sums2, int64_t n_blocks)
{
    const int32_t *deg = deg2 + blockIdx.y * stride;
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) sum += i0 + k < n ? deg[i0 + k] : 0;
    int total;
    (void)block_scan_incl(sum, &total);
    if (threadIdx.x == 0) sums2[blockIdx.y * (n_blocks + 1) + blockIdx.x] = total;
}

// step 3: every block rescans its counts from its offset (sums2 holds the exclusive scan of the block sums by now)
__global__ __launch_bounds__(kBlock) void k_csr_scan_blocks(const int32_t *__restrict__ deg2, int64_t n, int64_t stride,
                                                            const int32_t *__restrict__ sums2, int64_t n_blocks,
                                                            int32_t *__restrict__ in_ptr, int32_t *__restrict__ out_ptr)
{
    const int32_t *deg = deg2 + blockIdx.y * stride;
    int32_t *ptr = blockIdx.y ? out_ptr : in_ptr;
    const int32_t *sums = sums2 + blockIdx.y * (n_blocks + 1);
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = i0 + k < n ? deg[i0 + k] : 0;
        sum += v[k];
    }
    int total;
    int run = sums[blockIdx.x] + block_scan_incl(sum, &total) - sum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (i0 + k < n) ptr[i0 + k] = run;
        run += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ptr[n] = sums[n_blocks];
}

__global__ __launch_bounds__(kBlock) void k_csr_fill(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                     int64_t n_hits, int64_t n_segments, const int32_t *__restrict__ in_ptr,
                                                     const int32_t *__restrict__ out_ptr, int32_t *__restrict__ deg_in,
                                                     int32_t *__restrict__ deg_out, int32_t *__restrict__ tmp_in,
                                                     int32_t *__restrict__ tmp_out)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_segments) return;
    const int s = src[j], d = dst[j];
    if (!seg_ok(s, d, n_hits)) return;
    tmp_in[in_ptr[d] + atomicSub(deg_in + d, 1) - 1] = (int)j;
    tmp_out[out_ptr[s] + atomicSub(deg_out + s, 1) - 1] = (int)j;
}

// ---- batches of many segments: LDS-private counters per 16 k consecutive segments -------------------------------
constexpr int kWgSegs = 16384, kWgRange = 16384, kWgThreads = 1024, kWgPer = kWgSegs / kWgThreads;

// the hit range [lo, hi] the valid segments [b0, b1) touch; every thread gets it (red: 32 ints of LDS)
__device__ __forceinline__ void wg_range(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, int64_t b0,
                                         int64_t b1, int64_t n_hits, int *red, int &lo, int &hi, int &bad)
{
    int mn = 0x7FFFFFFF, mx = -1;
    for (int64_t j = b0 + threadIdx.x; j < b1; j += kWgThreads) {
        const int s = src[j], d = dst[j];
        if (seg_ok(s, d, n_hits)) {
            mn = min(mn, min(s, d));
            mx = max(mx, max(s, d));
        } else if (!(s < 0 && d < 0)) bad = 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = mn; red[2 * (threadIdx.x >> 6) + 1] = mx; }
    __syncthreads();
    lo = red[0]; hi = red[1];
#pragma unroll
    for (int w = 1; w < kWgThreads / 64; ++w) {
        lo = min(lo, red[2 * w]);
        hi = max(hi, red[2 * w + 1]);
    }
}

__global__ __launch_bounds__(kWgThreads) void k_csr_degrees_wg(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                              int64_t n_hits, int64_t n_segments, int32_t *deg_in,
                                                              int32_t *deg_out, int32_t *status)
{
    __shared__ int cin[kWgRange], cout[kWgRange];
    __shared__ int red[2 * (kWgThreads / 64)];
    const int64_t b0 = (int64_t)blockIdx.x * kWgSegs;
    const int64_t b1 = min(b0 + kWgSegs, n_segments);
    int lo, hi, bad = 0;
    wg_range(src, dst, b0, b1, n_hits, red, lo, hi, bad);
    if (bad) atomicOr(status, 1);
    if (hi < lo) return;                                 // no valid segment here
    const bool local = hi - lo < kWgRange;
    if (local)
        for (int i = threadIdx.x; i <= hi - lo; i += kWgThreads) cin[i] = cout[i] = 0;
    __syncthreads();
    for (int64_t j = b0 + threadIdx.x; j < b1; j += kWgThreads) {
        const int s = src[j], d = dst[j];
        if (!seg_ok(s, d, n_hits)) continue;
        if (local) {
            atomicAdd(&cin[d - lo], 1);
            atomicAdd(&cout[s - lo], 1);
        } else {
            atomicAdd(deg_in + d, 1);
            atomicAdd(deg_out + s, 1);
        }
    }
    if (!local) return;
    __syncthreads();
    for (int i = threadIdx.x; i <= hi - lo; i += kWgThreads) {
        const int a = cin[i], b = cout[i];
        if (a) atomicAdd(deg_in + lo + i, a);
        if (b) atomicAdd(deg_out + lo + i, b);
    }
}

// slots: a segment's LDS rank inside its workgroup's share of a list + the share's base, claimed once per touched hit
__global__ __launch_bounds__(kWgThreads) void k_csr_fill_wg(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                           int64_t n_hits, int64_t n_segments,
                                                           const int32_t *__restrict__ in_ptr,
                                                           const int32_t *__restrict__ out_ptr, int32_t *deg_in,
                                                           int32_t *deg_out, int32_t *__restrict__ tmp_in,
                                                           int32_t *__restrict__ tmp_out)
{
    __shared__ int cin[kWgRange], cout[kWgRange];
    __shared__ int red[2 * (kWgThreads / 64)];
    const int64_t b0 = (int64_t)blockIdx.x * kWgSegs;
    const int64_t b1 = min(b0 + kWgSegs, n_segments);
    int lo, hi, bad = 0;
    wg_range(src, dst, b0, b1, n_hits, red, lo, hi, bad);
    if (hi < lo) return;
    if (hi - lo >= kWgRange) {                           // endpoints all over the batch: a global claim per segment
        for (int64_t j = b0 + threadIdx.x; j < b1; j += kWgThreads) {
            const int s = src[j], d = dst[j];
            if (!seg_ok(s, d, n_hits)) continue;
            tmp_in[in_ptr[d] + atomicSub(deg_in + d, 1) - 1] = (int)j;
            tmp_out[out_ptr[s] + atomicSub(deg_out + s, 1) - 1] = (int)j;
        }
        return;
    }
    for (int i = threadIdx.x; i <= hi - lo; i += kWgThreads) cin[i] = cout[i] = 0;
    __syncthreads();
    int ss[kWgPer], dd[kWgPer], ri[kWgPer], ro[kWgPer];
#pragma unroll
    for (int k = 0; k < kWgPer; ++k) {
        const int64_t j = b0 + threadIdx.x + (int64_t)k * kWgThreads;
        ss[k] = dd[k] = -1;
        if (j < b1) { ss[k] = src[j]; dd[k] = dst[j]; }
        if (!seg_ok(ss[k], dd[k], n_hits)) ss[k] = -1;
        if (ss[k] >= 0) {
            ri[k] = atomicAdd(&cin[dd[k] - lo], 1);
            ro[k] = atomicAdd(&cout[ss[k] - lo], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= hi - lo; i += kWgThreads) {
        const int a = cin[i], b = cout[i];
        if (a) cin[i] = in_ptr[lo + i] + atomicSub(deg_in + lo + i, a) - a;
        if (b) cout[i] = out_ptr[lo + i] + atomicSub(deg_out + lo + i, b) - b;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWgPer; ++k) {
        if (ss[k] < 0) continue;
        const int j = (int)(b0 + threadIdx.x + (int64_t)k * kWgThreads);
        tmp_in[cin[dd[k] - lo] + ri[k]] = j;
        tmp_out[cout[ss[k] - lo] + ro[k]] = j;
    }
}

__device__ __forceinline__ int rank_in(const int32_t *__restrict__ lst, int b, int e, int j)
{
    int r = 0;
    int k = b;
    for (; k + 4 <= e; k += 4) {                       // four independent loads in flight
        const int a0 = lst[k], a1 = lst[k + 1], a2 = lst[k + 2], a3 = lst[k + 3];
        r += (a0 < j) + (a1 < j) + (a2 < j) + (a3 < j);
    }
    for (; k < e; ++k) r += lst[k] < j;
    return r;
}

__global__ __launch_bounds__(kBlock) void k_csr_rank(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                     int64_t n_hits, int64_t n_segments, const int32_t *__restrict__ in_ptr,
                                                     const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ tmp_in,
                                                     const int32_t *__restrict__ tmp_out, int32_t *__restrict__ in_eid,
                                                     int32_t *__restrict__ in_nbr, int32_t *__restrict__ out_eid,
                                                     int32_t *__restrict__ out_nbr, const int32_t *__restrict__ bad_word,
                                                     int32_t *__restrict__ status)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_segments) return;
    if (j == 0) *status = *bad_word;                   // (the degrees pass collected it next to the counts: one memset)
    if (j >= in_ptr[n_hits]) {                         // the tail past the valid entries: defined, never walked
        in_eid[j] = in_nbr[j] = out_eid[j] = out_nbr[j] = -1;
    }
    const int s = src[j], d = dst[j];
    if (!seg_ok(s, d, n_hits)) return;
    {
        const int b = in_ptr[d];
        const int slot = b + rank_in(tmp_in, b, in_ptr[d + 1], (int)j);
        in_eid[slot] = (int)j;
        in_nbr[slot] = s;
    }
    {
        const int b = out_ptr[s];
        const int slot = b + rank_in(tmp_out, b, out_ptr[s + 1], (int)j);
        out_eid[slot] = (int)j;
        out_nbr[slot] = d;
    }
}

struct CsrWs {
    int32_t *bad;            // the malformed-segment flag of the degrees pass (copied to the caller's status at the end)
    int32_t *deg;            // [2][stride]  counts, then countdown cursors
    int32_t *tmp_in, *tmp_out;   // [n_segments] each
    int32_t *sums;           // [2][n_blocks + 1]
    int64_t stride, n_blocks;
    size_t bytes;
};

CsrWs carve_csr(char *base, int64_t n_hits, int64_t n_segments)
{
    CsrWs w;
    w.stride = (n_hits + 63) & ~(int64_t)63;
    w.n_blocks = (n_hits + kScanTile - 1) / kScanTile;
    size_t off = 0;
    auto take = [&](size_t n) { char *p = base ? base + off : nullptr; off += align256(n); return p; };
    // [bad word + 63 pad | counts in | counts out]: cleared by ONE memset
    w.bad = reinterpret_cast<int32_t *>(take((size_t)(64 + 2 * w.stride) * sizeof(int32_t)));
    w.deg = w.bad ? w.bad + 64 : nullptr;
    w.tmp_in = reinterpret_cast<int32_t *>(take((size_t)n_segments * sizeof(int32_t)));
    w.tmp_out = reinterpret_cast<int32_t *>(take((size_t)n_segments * sizeof(int32_t)));
    w.sums = reinterpret_cast<int32_t *>(take((size_t)2 * (w.n_blocks + 1) * sizeof(int32_t)));
    w.bytes = off + 256;
    return w;
}

}  // namespace
}  // namespace gnn

using namespace gnn;

extern "C" {

size_t gnn_csr_build_workspace_bytes(int64_t n_hits, int64_t n_segments)
{
    if (n_hits < 0 || n_segments < 0) return 0;
    return carve_csr(nullptr, n_hits, n_segments).bytes;
}

int gnn_csr_build(const int32_t *src, const int32_t *dst, int64_t n_hits, int64_t n_segments, int32_t *in_ptr,
                  int32_t *in_eid, int32_t *in_nbr, int32_t *out_ptr, int32_t *out_eid, int32_t *out_nbr,
                  int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    ProfChain chain_;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_hits < 0 || n_segments < 0 || n_hits >= (int64_t)1 << 31 || n_segments >= (int64_t)1 << 31)
        return fail(GNN_ERR_BADARG, "gnn_csr_build: sizes outside the int32 index range");
    if (!in_ptr || !out_ptr || !status)
        return fail(GNN_ERR_BADARG, "gnn_csr_build: output pointer missing");
    if (n_segments > 0 && (!src || !dst || !in_eid || !in_nbr || !out_eid || !out_nbr))
        return fail(GNN_ERR_BADARG, "gnn_csr_build: segment array missing");
    const size_t need = gnn_csr_build_workspace_bytes(n_hits, n_segments);
    if (!workspace || workspace_bytes < need) return fail(GNN_ERR_WORKSPACE, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    CsrWs w = carve_csr(base, n_hits, n_segments);
    hipError_t err = hipMemsetAsync(n_segments > 0 ? w.bad : status, 0,
                                    n_segments > 0 ? (size_t)(64 + 2 * w.stride) * sizeof(int32_t) : sizeof(int32_t), s);
    if (err == hipSuccess && n_segments == 0 && w.stride > 0)
        err = hipMemsetAsync(w.deg, 0, (size_t)2 * w.stride * sizeof(int32_t), s);
    if (err != hipSuccess) return fail(-(int)err, "gnn_csr_build: memset failed: %s", hipGetErrorString(err));
    const unsigned gseg = (unsigned)((n_segments + kBlock - 1) / kBlock);
    const bool big = n_segments >= kWgMinSegments && !getenv("GNN_CSR_NO_LDS");
    const unsigned gwg = (unsigned)((n_segments + kWgSegs - 1) / kWgSegs);
    if (n_segments > 0 && big)
        GNN_LAUNCH("k_csr_degrees_wg", k_csr_degrees_wg, gwg, kWgThreads, s, src, dst, n_hits, n_segments, w.deg,
                   w.deg + w.stride, w.bad);
    else if (n_segments > 0)
        GNN_LAUNCH("k_csr_degrees", k_csr_degrees, gseg, kBlock, s, src, dst, n_hits, n_segments, w.deg, w.deg + w.stride,
                   w.bad);
    if (n_hits <= kOneBlockMax) {
        GNN_LAUNCH("k_csr_scan", k_csr_scan_one, 2, kScanOneThreads, s, w.deg, in_ptr, out_ptr, n_hits, w.stride);
    } else {
        const dim3 g((unsigned)w.n_blocks, 2);
        GNN_LAUNCH("k_csr_scan_sums", k_csr_scan_sums, g, kBlock, s, w.deg, n_hits, w.stride, w.sums, w.n_blocks);
        // the block sums of both arrays, scanned in place ([n_blocks + 1] each: the last entry = the total)
        GNN_LAUNCH("k_csr_scan", k_csr_scan_one, 2, kScanOneThreads, s, w.sums, w.sums, w.sums + (w.n_blocks + 1), w.n_blocks,
                   w.n_blocks + 1);
        GNN_LAUNCH("k_csr_scan_blocks", k_csr_scan_blocks, g, kBlock, s, w.deg, n_hits, w.stride, w.sums, w.n_blocks, in_ptr,
                   out_ptr);
    }
    if (n_segments > 0) {
        if (big)
            GNN_LAUNCH("k_csr_fill_wg", k_csr_fill_wg, gwg, kWgThreads, s, src, dst, n_hits, n_segments, in_ptr, out_ptr,
                       w.deg, w.deg + w.stride, w.tmp_in, w.tmp_out);
        else
            GNN_LAUNCH("k_csr_fill", k_csr_fill, gseg, kBlock, s, src, dst, n_hits, n_segments, in_ptr, out_ptr, w.deg,
                       w.deg + w.stride, w.tmp_in, w.tmp_out);
        GNN_LAUNCH("k_csr_rank", k_csr_rank, gseg, kBlock, s, src, dst, n_hits, n_segments, in_ptr, out_ptr, w.tmp_in,
                   w.tmp_out, in_eid, in_nbr, out_eid, out_nbr, w.bad, status);
    }
    return 0;
}

}  // extern "C"
