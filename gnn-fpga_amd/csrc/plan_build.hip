// plan_build.hip - the execution plan of the fused pipeline (gnn-fpga_amd/plan.py), built on the GPU.
//
// What it replaces: the reference builds every batch on the host - graph_from_sparse densifies each
// graph into two [N, E] matrices and merge_graphs zero-pads them (gnn/graph.py:28-35,
// gnn/trainSegmentClassifier.py:66-111).  The index-form counterpart of that per-batch work is the
// plan of plan.py (levels, tiles, windows, SELL-16 lists, wave schedules, edge chunks).  plan.py is
// the specification (numpy); plan_device.py runs the same recipe as ~300 torch ops (0.2 s for
// 25.6 M segments); this file is the same plan, ARRAY FOR ARRAY, as a few dozen HIP kernels:
//
//   degrees (atomics) -> topological levels (Jacobi sweeps with early exit, same fixpoint and the
//   same 64-sweep cap as plan.topological_levels) -> hits sorted by (graph, level, -in, -out)
//   [radix sort] -> (graph, level) units -> tiles (the sequential greedy packing, one thread,
//   units staged through LDS) -> degree sort inside tiles [radix sort] -> padded new ids ->
//   segments sorted by end / start hit [2 radix sorts of (key, other end) pairs: a hit's
//   neighbours come out contiguous, in ascending segment order] -> windows per tile -> SELL-16
//   slice lengths and offsets [one scan of 4-vectors] -> edge chunks (run detection, marks, equal
//   parts) and their windows -> sizes.              (gnn_plan_build_sizes, one host read-back)
//   Then, into arrays the caller allocates from those sizes: lists (32-bit and 16-bit packed),
//   tile / chunk descriptors, wave schedules (the two-phase greedy of plan.py in float64, one
//   wavefront per tile), renumbered X, endpoints.              (gnn_plan_build_fill)
//
// The recipe above is the GLOBAL form (gnn_plan_build_sizes): it treats a batch as one big graph.  A batch that names its
// graphs' segment ranges (gnn_plan_build_sizes_graphs, ABI 5) takes the GRAPH-LOCAL form of stage 1 instead - the same
// arrays, entry for entry, from kernels that keep one graph's (pb_graph_levels, pb_graph_renumber) or one tile's
// (pb_tile_sort, pb_tile_lists) tables in LDS: no level sweeps over all segments, no device-wide sort; see the block
// comments at those kernels.  It checks the layout it is told and reports status 128 when it does not hold.
//
// Device-wide radix sorts and scans are rocPRIM's (plain library primitives, like a library GEMM);
// everything else is hand-written.  All kernels are grid-stride on bounded grids (an early-exit
// sweep costs a launch, not a dispatch of 100 k idle workgroups).  Integer work: bit-exact against
// plan.py (tests/test_plan_hip.py compares every array).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {
using namespace gnn;

constexpr int TB = 256;
constexpr int SLICE = 16;
constexpr int kMaxLevelIters = 64;     // plan.topological_levels(max_iter=64)
constexpr int kMaxSlicesPerTile = 128; // tile_hits <= 2048

inline unsigned gs(int64_t n)
{
    int64_t g = (n + TB - 1) / TB;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
#define GS_LOOP(I, N) \
    for (int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; I < (N); I += (int64_t)gridDim.x * blockDim.x)

// device header (int64 slots) - counts that later kernels read and the final sizes are made from
enum Hdr {
    H_NVALID, H_NUNITS, H_NTILES, H_NPAD, H_TILEMAX, H_NSCHED, H_NRUNS, H_NMARKS, H_NCHUNKS, H_STATUS,
    H_LDSREC, H_NLDSTILES, H_LDSIN, H_LDSOUT, H_EDGEROWS, H_NLDSCHUNKS, H_MAXSTEPS, H_MAXLEVEL, H_LISTMODE, H_SPARSE, H_COUNT = 32
};
enum Status { ST_OK = 0, ST_DEGREE = 1, ST_TILES = 2, ST_PAD = 4, ST_CHUNKS = 8, ST_SLICES = 16, ST_I32 = 32,
              ST_ENDPOINT = 64 };   // ST_ENDPOINT: a segment end outside [0, n_hits), or exactly one end negative

struct I2 { int a, b; };
struct I4 { int a, b, c, d; };
struct PlusI2 { __host__ __device__ I2 operator()(const I2 &x, const I2 &y) const { return {x.a + y.a, x.b + y.b}; } };
struct PlusI4 {
    __host__ __device__ I4 operator()(const I4 &x, const I4 &y) const { return {x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d}; }
};

// "the last key that is not -1" as a scan operator (associative, not commutative): pb_seg_keys
struct CarryKey { __host__ __device__ int operator()(int a, int b) const { return b >= 0 ? b : a; } };

// ---- workspace ------------------------------------------------------------------------------------
struct Bounds { int64_t nt_max, np_max, ns_max, m_max, c_max; };
inline Bounds bounds_of(int64_t n, int64_t E, int CH)
{
    Bounds b;
    b.nt_max = n / 16 + 1024;               // average tile of >= 16 hits, else ST_TILES (caller falls back)
    b.np_max = n + 15 * b.nt_max;
    b.np_max = (b.np_max + 15) / 16 * 16;
    b.ns_max = b.np_max / 16;
    b.m_max = 8 * (E / (CH > 0 ? CH : 1)) + 16;        // marks: two per run of >= CH/4 segments
    b.c_max = 9 * (E / (CH > 0 ? CH : 1)) + 32;        // chunks: one per CH segments + one per mark
    return b;
}

struct Ws {
    int64_t *hdr;
    int *chg;                                  // [kMaxLevelIters] "this sweep raised a level"
    // hits
    int *deg_in, *deg_out, *gid, *lvA, *lvB, *down, *iota, *base, *oor, *tpos, *uflag, *uscan, *ustart, *inv, *hkey;
    unsigned long long *k64a, *k64b;
    // tiles
    int *tile_bounds, *tpad_off, *sbase, *t_desc;      // t_desc: [nt_max][8]
    // padded hits / slices
    I2 *degn, *ptr;                            // (in, out) degree and CSR pointer per new id
    I4 *sl4, *off4;                            // per slice: (16 steps_in, 16 steps_out, 16 ceil8 in, 16 ceil8 out), scanned
    int *slice_tile;                           // tile of a slice (pb_new_ids -> pb_graph_renumber, pb_fill_lists)
    int *trange;                               // [nt_max + 1][4] segment range of a tile's in / out lists (lo, -hi)
    // segments
    int *src_new, *dst_new, *kscr, *sv_in, *sv_out, *c_scan, *rb;
    int *mscan, *marks, *parts, *pscan, *cb, *c_desc;  // c_desc: [c_max][8]
    I2 *pin, *pout;                            // graph-local form: (segment, other end) pairs in list places
    void *temp;
    size_t temp_bytes;
    size_t bytes;
};

size_t rocprim_temp_bytes(int64_t n, int64_t E, const Bounds &b)
{
    size_t m = 0, t = 0;
    auto up = [&](size_t x) { if (x > m) m = x; };
    (void)rocprim::radix_sort_pairs(nullptr, t, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int *)nullptr, (int *)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0, false);
    up(t);
    (void)rocprim::radix_sort_pairs(nullptr, t, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                                    (int *)nullptr, (size_t)E, 0u, 32u, (hipStream_t)0, false);
    up(t);
    (void)rocprim::exclusive_scan(nullptr, t, (const int *)nullptr, (int *)nullptr, 0, (size_t)(E > n ? E : n) + 1,
                                  rocprim::plus<int>(), (hipStream_t)0, false);
    up(t);
    (void)rocprim::inclusive_scan(nullptr, t, (const int *)nullptr, (int *)nullptr, (size_t)E + 1,
                                  CarryKey(), (hipStream_t)0, false);
    up(t);
    (void)rocprim::exclusive_scan(nullptr, t, (const I2 *)nullptr, (I2 *)nullptr, I2{0, 0}, (size_t)b.np_max + 1, PlusI2(),
                                  (hipStream_t)0, false);
    up(t);
    (void)rocprim::exclusive_scan(nullptr, t, (const I4 *)nullptr, (I4 *)nullptr, I4{0, 0, 0, 0}, (size_t)b.ns_max + 1,
                                  PlusI4(), (hipStream_t)0, false);
    up(t);
    return m + 256;
}

Ws carve(char *p, int64_t n, int64_t E, int CH)
{
    const Bounds b = bounds_of(n, E, CH);
    Ws w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += align256(bytes);
        return q;
    };
    auto ints = [&](int64_t k) { return reinterpret_cast<int *>(take((size_t)k * sizeof(int))); };
    w.hdr = reinterpret_cast<int64_t *>(take(H_COUNT * sizeof(int64_t)));
    w.chg = ints(kMaxLevelIters + 2);
    w.deg_in = ints(n); w.deg_out = ints(n);                    // (one memset clears hdr .. lvB)
    w.lvA = ints(n); w.lvB = ints(n);
    w.gid = ints(n); w.down = ints(n); w.hkey = ints(n);
    w.iota = ints(n); w.base = ints(n); w.oor = ints(n); w.tpos = ints(n); w.uflag = ints(n + 1); w.uscan = ints(n + 1);
    w.ustart = ints(n + 2); w.inv = ints(n + 1);
    w.k64a = reinterpret_cast<unsigned long long *>(take((size_t)n * 8));
    w.k64b = reinterpret_cast<unsigned long long *>(take((size_t)n * 8));
    w.tile_bounds = ints(b.nt_max + 2); w.tpad_off = ints(b.nt_max + 2); w.sbase = ints(b.nt_max + 2);
    w.t_desc = ints((b.nt_max + 1) * 8);
    w.degn = reinterpret_cast<I2 *>(take((size_t)(b.np_max + 1) * sizeof(I2)));
    w.ptr = reinterpret_cast<I2 *>(take((size_t)(b.np_max + 1) * sizeof(I2)));
    w.sl4 = reinterpret_cast<I4 *>(take((size_t)(b.ns_max + 1) * sizeof(I4)));
    w.off4 = reinterpret_cast<I4 *>(take((size_t)(b.ns_max + 1) * sizeof(I4)));
    w.slice_tile = ints(b.ns_max + 1);
    w.trange = ints((b.nt_max + 1) * 4);
    w.src_new = ints(E); w.dst_new = ints(E); w.kscr = ints(E + 1); w.sv_in = ints(E); w.sv_out = ints(E);
    w.c_scan = ints(E + 1); w.rb = ints(E + 2);
    w.mscan = ints(E + 1);
    w.marks = ints(b.m_max + 2); w.parts = ints(b.m_max + 2); w.pscan = ints(b.m_max + 2);
    w.cb = ints(b.c_max + 2); w.c_desc = ints((b.c_max + 1) * 8);
    w.pin = reinterpret_cast<I2 *>(take((size_t)E * sizeof(I2)));
    w.pout = reinterpret_cast<I2 *>(take((size_t)E * sizeof(I2)));
    w.temp_bytes = rocprim_temp_bytes(n, E, b);
    w.temp = take(w.temp_bytes);
    w.bytes = off;
    return w;
}

__device__ __forceinline__ void set_status(int64_t *hdr, int bit)
{
    atomicOr(reinterpret_cast<unsigned long long *>(hdr + H_STATUS), (unsigned long long)bit);
}
__device__ __forceinline__ void hdr_max(int64_t *hdr, int slot, long long v)     // v >= 0
{
    atomicMax(reinterpret_cast<unsigned long long *>(hdr + slot), (unsigned long long)v);
}
__device__ __forceinline__ void hdr_add(int64_t *hdr, int slot, long long v)
{
    atomicAdd(reinterpret_cast<unsigned long long *>(hdr + slot), (unsigned long long)v);
}
// One atomic per WORKGROUP on a header word, not one per wave or per thread: tens of thousands of
// atomics on ONE address serialise at the memory side (pb_degrees' valid count and pb_fill_hits' three
// feature maxima were most of those kernels' time at c3 x 256).  All threads of the workgroup call.
template <typename OP>
__device__ __forceinline__ int block_reduce_i(int v, int *red, OP op)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = op(r, red[w]);
    return r;
}
struct OpAdd { __device__ int operator()(int a, int b) const { return a + b; } };
struct OpMax { __device__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct OpOr { __device__ int operator()(int a, int b) const { return a | b; } };

// ---- stage 1 kernels ------------------------------------------------------------------------------
// In / out degree of every hit.  Two global atomics per segment (51 M at c3 x 256) were 2 ms of a
// 10 ms plan.  Segments arrive grouped (a graph's segments are contiguous, the reference emits them per
// layer pair): a workgroup takes 16 k consecutive segments, finds the hit range their endpoints fall
// into and, when it is narrow enough, counts in LDS and flushes only the touched counters - a few
// thousand global atomics per workgroup instead of 32 k.  Wide ranges (shuffled input) count in global
// memory as before.
constexpr int kDegSegs = 16384, kDegRange = 16384, kDegWindows = 4;
// The builder does not trust its endpoints: a segment is VALID when both ends lie in [0, n); a padded
// segment has both ends negative; anything else (an end >= n, one end negative) sets ST_ENDPOINT and
// is skipped by every kernel that indexes per-hit arrays before the caller reads the status back.
__device__ __forceinline__ bool seg_ok(int s, int d, int n) { return (unsigned)s < (unsigned)n && (unsigned)d < (unsigned)n; }

__global__ __launch_bounds__(1024) void pb_degrees(const int *__restrict__ src, const int *__restrict__ dst, int64_t E, int n,
                                                  int *deg_in, int *deg_out, int64_t *hdr)
{
    __shared__ int cin[kDegRange], cout[kDegRange];
    __shared__ int red[2 * 16];
    int cnt = 0, bad = 0;
    for (int64_t b0 = (int64_t)blockIdx.x * kDegSegs; b0 < E; b0 += (int64_t)gridDim.x * kDegSegs) {
        const int64_t b1 = b0 + kDegSegs < E ? b0 + kDegSegs : E;
        int mn = 0x7FFFFFFF, mx = -1;
        for (int64_t j = b0 + threadIdx.x; j < b1; j += 1024) {
            const int s = src[j], d = dst[j];
            if (seg_ok(s, d, n)) {
                mn = s < mn ? s : mn; mn = d < mn ? d : mn;
                mx = s > mx ? s : mx; mx = d > mx ? d : mx;
            } else if (s >= 0 || d >= 0) bad = 1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = mn; red[2 * (threadIdx.x >> 6) + 1] = mx; }
        __syncthreads();
        int lo = red[0], hi = red[1];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            lo = red[2 * w] < lo ? red[2 * w] : lo;
            hi = red[2 * w + 1] > hi ? red[2 * w + 1] : hi;
        }
        // up to kDegWindows windows of kDegRange hits, one pass over the block's segments (L2-resident by now) per
        // window: the 50 k-hit graphs of the mu200 shape number their hits in no layer order, so a block's ends span
        // the whole graph - two global atomics per segment cost 0.31 ms of that batch's 1.7 ms plan
        const int span = hi >= lo ? hi - lo + 1 : 0;
        const int nwin = (span + kDegRange - 1) / kDegRange;
        const bool local = nwin >= 1 && nwin <= kDegWindows;
        if (!local) {
            for (int64_t j = b0 + threadIdx.x; j < b1; j += 1024) {
                const int s = src[j], d = dst[j];
                if (seg_ok(s, d, n)) {
                    atomicAdd(&deg_out[s], 1);
                    atomicAdd(&deg_in[d], 1);
                    ++cnt;
                }
            }
            __syncthreads();
            continue;                                      // (workgroup-uniform)
        }
        for (int w = 0; w < nwin; ++w) {
            const int wlo = lo + w * kDegRange, wn = span - w * kDegRange < kDegRange ? span - w * kDegRange : kDegRange;
            for (int i = threadIdx.x; i < wn; i += 1024) cin[i] = cout[i] = 0;
            __syncthreads();
            for (int64_t j = b0 + threadIdx.x; j < b1; j += 1024) {
                const int s = src[j], d = dst[j];
                if (seg_ok(s, d, n)) {
                    if ((unsigned)(s - wlo) < (unsigned)wn) atomicAdd(&cout[s - wlo], 1);
                    if ((unsigned)(d - wlo) < (unsigned)wn) atomicAdd(&cin[d - wlo], 1);
                    if (w == 0) ++cnt;
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < wn; i += 1024) {
                const int a = cin[i], b = cout[i];
                if (a) atomicAdd(&deg_in[wlo + i], a);
                if (b) atomicAdd(&deg_out[wlo + i], b);
            }
            __syncthreads();
        }
    }
    cnt = block_reduce_i(cnt, red, OpAdd());
    bad = block_reduce_i(bad, red, OpOr());
    if (threadIdx.x == 0 && cnt) hdr_add(hdr, H_NVALID, cnt);
    if (threadIdx.x == 0 && bad) set_status(hdr, ST_ENDPOINT);
}

// gid[i] = number of interior graph boundaries hit_ptr[1..G-1] that are <= i   (plan.py: add.at + cumsum)
__global__ __launch_bounds__(TB) void pb_gid(const int64_t *__restrict__ hit_ptr, int64_t G, int64_t n, int *gid)
{
    GS_LOOP(i, n) {
        int64_t lo = 1, hi = G;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (hit_ptr[mid] <= i) lo = mid + 1; else hi = mid;
        }
        gid[i] = (int)(lo - 1);
    }
}

// Sweep t (1-based) of the Jacobi longest-path relaxation of plan.topological_levels, IN PLACE and
// without atomics.  From an all-zero start the Jacobi iterate is level_t[v] = min(t, longest walk
// ending at v): sweep t changes exactly the hits that go from t-1 to t, those with a segment from a
// hit that stood at t-1 - and a hit below t-1 at the start of sweep t never changes again.  So
// "level[s] >= t-1 -> level[d] = t" on ONE buffer is the Jacobi sweep: every writer stores the same
// value, and a reader that sees a start hit already raised to t draws the same conclusion as one
// that sees t-1.  (Values only come from earlier launches or are this very t: no stale-cache case.)
// A sweep that stored nothing (chg[t] == 0) found the fixpoint; all later sweeps return at once.
// Levels are swept as BYTES (the iterate never exceeds kMaxLevelIters = 64): 2.5 MB for the 2.56 M hits of
// c3 x 256, which every XCD's 4 MB L2 holds, where the int32 table's 10 MB lived in the MALL - the
// sweep is one random level[src] read per segment and nothing else.  pb_widen hands int32 levels on.
// BLOCK LIVENESS: the segments that raise a hit in sweep t are exactly those whose start hit has a
// longest walk >= t - 1 - a set that only shrinks as t grows.  So a block of consecutive
// segments that raised nothing in one sweep never raises anything again and is skipped from then on
// (one byte per block of `blk` segments).  The reference emits segments grouped by layer pair (gnn/graph.py:80-93): the
// block of layer pair (l, l + 1) dies after sweep l + 1, and the 12 sweeps of a 10-layer batch read
// about half of what 12 full passes over src / dst read.  Any segment order stays exact.
// (blocks of `blk` segments: 4096 for large batches, 1024 for small ones so that a single graph still
// spreads over ~100 workgroups; eight segments per thread are in flight at a time)
__global__ __launch_bounds__(TB) void pb_level_sweep(const int *__restrict__ src, const int *__restrict__ dst, int64_t E, int n,
                                                     unsigned char *level, int *chg, int t, unsigned char *dead, int blk)
{
    if (t > 1 && chg[t - 1] == 0) return;
    const int64_t nb = (E + blk - 1) / blk;
    int fired = 0;
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        if (dead && dead[b]) continue;                     // (workgroup-uniform)
        const int64_t j0 = b * blk, j1 = j0 + blk < E ? j0 + blk : E;
        int any = 0;
        for (int64_t jb = j0 + threadIdx.x; jb < j1; jb += 8 * TB) {
            int sv[8], dv[8], lv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t j = jb + (int64_t)u * TB;
                sv[u] = j < j1 ? src[j] : -1;
                dv[u] = j < j1 ? dst[j] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) lv[u] = seg_ok(sv[u], dv[u], n) ? (int)level[sv[u]] : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lv[u] >= t - 1) {                      // (t >= 1: never true for the -1 of an invalid segment)
                    level[dv[u]] = (unsigned char)t;
                    any = 1;
                }
        }
        any = __syncthreads_or(any);
        if (!any && dead && threadIdx.x == 0) dead[b] = 1;
        fired |= any;
    }
    if (fired && threadIdx.x == 0 && chg[t] == 0) chg[t] = 1;
}

__global__ __launch_bounds__(TB) void pb_widen(const unsigned char *__restrict__ a, int *b, int64_t n)
{
    GS_LOOP(i, n) b[i] = a[i];
}

// (One cooperative launch with grid.sync() between sweeps was measured and dropped: a grid barrier
// with its cross-XCD L2 write-back costs 45-160 us per sweep on this part, against 5 us for an empty
// launch and 20-210 us for a full one - 0.51 ms instead of 0.42 for one detector graph, 3.1 ms
// instead of 2.3 for c3 x 256.)
// Small batches: all sweeps in ONE launch by one workgroup (its waves share the CU's L1, so a
// workgroup barrier orders the stores of sweep t before the loads of sweep t + 1) - 64 launches
// would cost more than the sweeps themselves.
constexpr int kSweepRound = 12;
constexpr int64_t kSmallSweepSegments = 1 << 15;   // (100 k segments: 0.69 ms in one workgroup, 0.41 ms as 64 launches)
__global__ __launch_bounds__(1024) void pb_levels_small(const int *__restrict__ src, const int *__restrict__ dst, int E, int n,
                                                        int *level)
{
    for (int t = 1; t <= kMaxLevelIters; ++t) {
        int any = 0;
        for (int j = threadIdx.x; j < E; j += 1024) {
            const int s = src[j], d = dst[j];
            if (seg_ok(s, d, n) && level[s] >= t - 1) {
                level[d] = t;
                any = 1;
            }
        }
        if (!__syncthreads_or(any)) break;
    }
}

// ---- graph-local stage 1 (ABI 5: gnn_plan_build_sizes_graphs) -------------------------------------------
// When the caller names the graphs' segment ranges (seg_ptr: graph g owns segments [seg_ptr[g], seg_ptr[g+1]) and
// they join hits of [hit_ptr[g], hit_ptr[g+1]) only - what merge_graphs' block-diagonal batches are,
// gnn/trainSegmentClassifier.py:66-95), degrees, levels and the first sort key are ONE launch: a workgroup per
// graph keeps the graph's levels and degree counters in LDS (8 bytes per hit) and walks the graph's segments a few
// times - where the global form above walks ALL segments once per level (12 launches for a 10-layer detector) after
// a pass of global atomics for the degrees: 1.6 of the 5.7 ms of a c3 x 256 plan.
//   round 1: degrees (LDS atomics; in | out counts share a word, a half that would wrap reports FAST_MISS) and a
//            first asynchronous relaxation  level[d] = max(level[d], level[s] + 1)  (LDS atomicMax);
//   round r >= 2: the same relaxation until a round raises nothing.  The fixpoint of the asynchronous iteration is
//            the longest walk ending at each hit - the Jacobi iterate of plan.topological_levels once it has
//            converged, whatever the order of the updates.  A level above kMaxLevelIters (a deeper graph or a
//            cycle: the capped Jacobi iterate is something else there) reports FAST_MISS.  Segments grouped by layer
//            pair in ascending order (gnn/graph.py:80-93) converge in round 1 or 2; one more round sees no change.
//   the "one below the nearest end hit" rule for hits without incoming segments (pb_down / pb_key1) rides in the
//            same rounds: such a hit's level is 0 by definition, so its LDS word is free to hold
//            min(level[d]) over its outgoing segments; it is reset before every round and only the last round's
//            value - taken from levels that no longer moved - is used.
// The workgroup checks what it was told: a valid segment with an end outside its graph's hit range, hit / segment
// ranges that do not tile [0, n) / [0, E), a graph larger than the LDS capacity: FAST_MISS, and the caller runs the
// global form (every output of this kernel is rewritten there).
constexpr int ST_FAST_MISS = 128;
constexpr int kGraphLdsBytes = 152 * 1024;
constexpr int kGraphCapHits = kGraphLdsBytes / 8;          // 19456 hits per graph: 8 bytes of tables, then the sort keys
constexpr int kGraphCapHits64 = 16384;                     // ... 8-byte keys up to here, beyond it only 4-byte ones fit
__host__ __device__ inline int pow2_ceil(int v) { int p = 2; while (p < v) p <<= 1; return p; }

// bitonic sort of M (a power of two) 64-bit words in LDS, ascending; every thread of the workgroup calls.
// A step costs an LDS round trip whatever the number of compare-exchanges, and a workgroup barrier on top of it
// costs several times that (16 waves: ~0.4 us per step, 105 steps for 16 k words).  Wave w therefore owns the pairs
// [w PW, (w + 1) PW) of a step: for strides j <= PW both words of its pairs lie in its own 2 PW words, and such steps
// need no workgroup barrier - the wave's LDS operations complete in order (a wave-level fence keeps the compiler from
// moving them).  Only the strides above PW (10 of the 105 steps at 16 k words and 16 waves) meet at __syncthreads().
template <typename W>
__device__ __forceinline__ void bitonic_sort_lds(W *a, int M)
{
    const int nw = (int)(blockDim.x >> 6), w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int PW = (M >> 1) / nw;                       // pairs per wave (0 for tiny M: every step at the barrier)
    for (int k = 2; k <= M; k <<= 1) {
        __syncthreads();                                // (the wave-local steps of the k before)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool local = PW >= 64 && j <= PW;
            const int i0 = local ? w * PW + lane : (int)threadIdx.x, i1 = local ? (w + 1) * PW : (M >> 1);
            const int st = local ? 64 : (int)blockDim.x;
            for (int i = i0; i < i1; i += st) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const W x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
            if (local) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }
        }
    }
    __syncthreads();
}
constexpr int kFastMaxDegree = 1024;                       // pb_sort_lists ranks a list in O(deg^2 / 16)

__global__ __launch_bounds__(1024) void pb_graph_levels(const int *__restrict__ src, const int *__restrict__ dst,
                                                        const int64_t *__restrict__ hit_ptr,
                                                        const int64_t *__restrict__ seg_ptr, int64_t G, int64_t n, int64_t E,
                                                        int cap, int *deg_in, int *deg_out, int *gid, int *level,
                                                        unsigned long long *key, int *iota, int *hkey, int *run_flag,
                                                        int *base, int *uflag, int64_t *hdr)
{
    extern __shared__ int lds[];
    int *lvl = lds;                                              // [cap]
    unsigned *dio = reinterpret_cast<unsigned *>(lds + cap);     // [cap]  in-degree | out-degree << 16
    __shared__ int red[16];
    __shared__ int gm[3];                                        // this graph's largest level, in- and out-degree
    const int B = (int)blockDim.x;
    int cnt = 0, bad = 0, miss = 0, lmax = 0;
    for (int64_t g = blockIdx.x; g < G; g += gridDim.x) {
        const int64_t h0 = hit_ptr[g], h1 = hit_ptr[g + 1], e0 = seg_ptr[g], e1 = seg_ptr[g + 1];
        bool ok = h0 <= h1 && e0 <= e1 && h1 - h0 <= cap && h0 >= 0 && h1 <= n && e0 >= 0 && e1 <= E;
        if (g == 0 && (h0 != 0 || e0 != 0)) ok = false;
        if (g == G - 1 && (h1 != n || e1 != E)) ok = false;
        if (!ok) { miss = 1; continue; }                         // (workgroup-uniform)
        const int nh = (int)(h1 - h0), lo = (int)h0, hi = (int)h1;
        __syncthreads();
        for (int i = threadIdx.x; i < nh; i += B) { lvl[i] = 0; dio[i] = 0u; }
        __syncthreads();
        // (four segments per thread in flight and the next four requested before these are worked on: one
        // workgroup per CU has only its own loads to hide their latency)
        auto load4 = [&](int64_t jb, int *sv, int *dv) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = jb + (int64_t)u * B;
                sv[u] = j < e1 ? src[j] : -1;
                dv[u] = j < e1 ? dst[j] : -1;
            }
        };
        int sv[4], dv[4];
        load4(e0 + threadIdx.x, sv, dv);
        for (int64_t jb = e0 + threadIdx.x; jb < e1; jb += 4 * (int64_t)B) {
            int sn_[4], dn_[4];
            load4(jb + 4 * (int64_t)B, sn_, dn_);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = sv[u], d = dv[u];
                if (s < 0 && d < 0) continue;
                if (!seg_ok(s, d, (int)n)) { bad = 1; continue; }
                if (s < lo || s >= hi || d < lo || d >= hi) { miss = 1; continue; }
                ++cnt;
                const unsigned a = atomicAdd(&dio[d - lo], 1u), b = atomicAdd(&dio[s - lo], 0x10000u);
                if ((a & 0xFFFFu) == 0xFFFFu || (b >> 16) == 0xFFFFu) miss = 1;
                atomicMax(&lvl[d - lo], lvl[s - lo] + 1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sv[u] = sn_[u]; dv[u] = dn_[u]; }
        }
        __syncthreads();
        int round = 2;
        bool over = false;
        for (; round <= kMaxLevelIters + 2; ++round) {
            for (int i = threadIdx.x; i < nh; i += B) {
                const unsigned w = dio[i];
                if ((w & 0xFFFFu) == 0u && (w >> 16) != 0u) lvl[i] = 0x7FFFFFFF;
            }
            __syncthreads();
            int changed = 0, deep = 0;
            load4(e0 + threadIdx.x, sv, dv);
            for (int64_t jb = e0 + threadIdx.x; jb < e1; jb += 4 * (int64_t)B) {
                int sn_[4], dn_[4];
                load4(jb + 4 * (int64_t)B, sn_, dn_);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = sv[u], d = dv[u];
                    if (s < lo || s >= hi || d < lo || d >= hi) continue;
                    const bool root = (dio[s - lo] & 0xFFFFu) == 0u;
                    const int v = (root ? 0 : lvl[s - lo]) + 1;
                    const int old = atomicMax(&lvl[d - lo], v);
                    if (old < v) { changed = 1; if (v > kMaxLevelIters) deep = 1; }
                    if (root) atomicMin(&lvl[s - lo], old > v ? old : v);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { sv[u] = sn_[u]; dv[u] = dn_[u]; }
            }
            over = __syncthreads_or(deep) != 0;
            if (over || !__syncthreads_or(changed)) break;
        }
        if (over || round > kMaxLevelIters + 2) { miss = 1; continue; }
        if (threadIdx.x < 3) gm[threadIdx.x] = 0;
        __syncthreads();
        int g_lv = 0, g_di = 0, g_do = 0;
        for (int i = threadIdx.x; i < nh; i += B) {
            const unsigned w = dio[i];
            const int di = (int)(w & 0xFFFFu), dout = (int)(w >> 16);
            int lv = lvl[i];
            if (di == 0) { lv = dout > 0 ? lv - 1 : 0; if (lv < 0) lv = 0; }
            if (di > kFastMaxDegree || dout > kFastMaxDegree) miss = 1;
            lmax = lv > lmax ? lv : lmax;
            g_lv = lv > g_lv ? lv : g_lv; g_di = di > g_di ? di : g_di; g_do = dout > g_do ? dout : g_do;
            const int h = lo + i;
            deg_in[h] = di; deg_out[h] = dout; level[h] = lv; gid[h] = (int)g; iota[h] = h;
            hkey[h] = (int)g * 128 + (lv & 127);
            key[h] = ((unsigned long long)(unsigned)g << 39) | ((unsigned long long)(lv & 127) << 32) |
                     ((unsigned long long)(0xFFFF - di) << 16) | (unsigned long long)(0xFFFF - dout);
            lvl[i] = lv;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int x = __shfl_xor(g_lv, o, 64), y = __shfl_xor(g_di, o, 64), z = __shfl_xor(g_do, o, 64);
            g_lv = x > g_lv ? x : g_lv; g_di = y > g_di ? y : g_di; g_do = z > g_do ? z : g_do;
        }
        if ((threadIdx.x & 63) == 0) { atomicMax(&gm[0], g_lv); atomicMax(&gm[1], g_di); atomicMax(&gm[2], g_do); }
        __syncthreads();
        // run starts of the (graph, start level) key for the edge chunks (pb_seg_keys + carry scan + pb_run_flags of
        // the global form): segment 0, and every valid segment whose start hit's level differs from that of the
        // valid segment before it - the first valid segment of a graph always does (its key names another graph).
        for (int64_t j = e0 + threadIdx.x; j < e1; j += B) {
            const int s = src[j];
            int f = 0;
            if (j == 0) f = 1;
            else if (s >= lo && s < hi) {
                int64_t q = j - 1;
                int ps = -1;
                for (; q >= e0; --q) { ps = src[q]; if (ps >= 0) break; }
                f = q < e0 ? 1 : ((ps >= lo && ps < hi) ? (lvl[ps - lo] != lvl[s - lo] ? 1 : 0) : 1);
            }
            run_flag[j] = f;
        }
        if (g == G - 1 && threadIdx.x == 0) { run_flag[E] = 0; uflag[n] = 0; }
        // The first hit sort of the global form - (graph, level, -in degree, -out degree), stable in the hit id - is
        // local to the graph: the keys this workgroup just wrote, with the hit's local id below them, sorted in LDS
        // (the level / degree tables are done with).  Rank r of the graph is rank hit_ptr[g] + r of the batch; a
        // (graph, level) unit starts where the level changes (pb_unit_flags).
        __syncthreads();
        if (nh > 0) {
            const int M = pow2_ceil(nh);
            // bits of this graph's levels, degrees and ids: 32-bit words when they fit (a detector graph: 4 + 5 + 5 +
            // 14) - the sort is bound by the LDS traffic of its steps, half of it with half the word
            auto bw = [](int v) { int b = 1; while ((1 << b) <= v) ++b; return b; };
            const int bi = bw(nh - 1), bo = bw(gm[2]), bd = bw(gm[1]), bl = bw(gm[0]);
            if (bl + bd + bo + bi <= 32) {                            // (workgroup-uniform)
                // in place: the key of hit i takes the word of lvl[i]; the padding words reach into the degree table, so
                // they are written once every key has been made
                unsigned *a = reinterpret_cast<unsigned *>(lds);
                const int mdi = gm[1], mdo = gm[2];
                for (int i = threadIdx.x; i < nh; i += B) {
                    const unsigned w = dio[i];
                    a[i] = ((unsigned)lvl[i] << (bd + bo + bi)) | ((unsigned)(mdi - (int)(w & 0xFFFFu)) << (bo + bi)) |
                           ((unsigned)(mdo - (int)(w >> 16)) << bi) | (unsigned)i;
                }
                __syncthreads();
                for (int i = nh + threadIdx.x; i < M; i += B) a[i] = 0xFFFFFFFFu;
                __syncthreads();
                bitonic_sort_lds(a, M);
                const unsigned idm = (1u << bi) - 1u;
                const int ls = bd + bo + bi;
                for (int r = threadIdx.x; r < nh; r += B) {
                    const unsigned e = a[r];
                    base[lo + r] = lo + (int)(e & idm);
                    uflag[lo + r] = (r == 0 || (a[r - 1] >> ls) != (e >> ls)) ? 1 : 0;
                }
            } else if (nh > kGraphCapHits64) {                        // (8-byte keys of > 16 k hits do not fit the LDS)
                miss = 1;
            } else {
                unsigned long long *a = reinterpret_cast<unsigned long long *>(lds);
                for (int i = threadIdx.x; i < M; i += B)
                    a[i] = i < nh ? ((key[lo + i] & ((1ull << 39) - 1ull)) << 15) | (unsigned long long)i : ~0ull;
                __syncthreads();
                bitonic_sort_lds(a, M);
                for (int r = threadIdx.x; r < nh; r += B) {
                    const unsigned long long e = a[r];
                    base[lo + r] = lo + (int)(e & 0x7FFFull);
                    uflag[lo + r] = (r == 0 || (a[r - 1] >> 47) != (e >> 47)) ? 1 : 0;
                }
            }
        }
    }
    cnt = block_reduce_i(cnt, red, OpAdd());
    bad = block_reduce_i(bad, red, OpOr());
    miss = block_reduce_i(miss, red, OpOr());
    lmax = block_reduce_i(lmax, red, OpMax());
    if (threadIdx.x == 0) {
        if (cnt) hdr_add(hdr, H_NVALID, cnt);
        if (bad) set_status(hdr, ST_ENDPOINT);
        if (miss) set_status(hdr, ST_FAST_MISS);
        hdr_max(hdr, H_MAXLEVEL, lmax);
    }
}

// Renumbered endpoints and both neighbour lists of a graph without a device-wide sort: the lists' places are known
// (ptr = scan of the degrees in new ids), a workgroup per graph keeps one cursor per hit in LDS (and the graph's new
// ids, when they fit) and drops (segment, other end) pairs into the lists in whatever order its atomics come;
// pb_sort_lists then puts every list into ascending segment order - what the two stable radix sorts of (end hit,
// other end) pairs delivered (1.2 ms + pb_renumber's 0.3 at c3 x 256; this pair of kernels: 0.6-0.77 + 0.37 ms).
// Status 0 here means every segment with src >= 0 is valid and graph-local (pb_graph_levels).
// One direction per pass: the 8-byte pairs land at random places of the lists of one detector level at a time (80 KB
// per direction for the 10 k segments of a layer pair); the streams bypass the cache (nontemporal), the second pass
// reads src / dst once more.  What bounds it (rocprofv3 TCC_EA0_WRREQ: 46 M write requests, 31 M of them partial, 1.9
// GB for 0.6 GB of payload): the L2 hands most pairs on as writes of their own.  Measured and dropped: every chunk of
// 4096 segments sorted in LDS by (hit, position) so that places come in segment order and no list needs sorting
// afterwards - bit-identical lists, but 78 barrier steps per chunk: 1.7-1.9 ms.
template <bool INV_LDS>
__global__ __launch_bounds__(1024) void pb_graph_lists(const int *__restrict__ src, const int *__restrict__ dst,
                                                       const int64_t *__restrict__ hit_ptr,
                                                       const int64_t *__restrict__ seg_ptr, int64_t G, int n, int cap,
                                                       const int *__restrict__ inv, const I2 *__restrict__ ptr, int *src_new,
                                                       int *dst_new, I2 *pin, I2 *pout, const int64_t *__restrict__ hdr)
{
    if (hdr[H_STATUS] || hdr[H_LISTMODE]) return;           // (LISTMODE 1: pb_tile_lists builds the lists)
    extern __shared__ int lds[];
    int *cin = lds, *cout = lds + cap, *linv = lds + 2 * cap;      // linv: new ids of the graph's hits (INV_LDS)
    const int B = (int)blockDim.x;
    const int npad = inv[n];
    for (int64_t g = blockIdx.x; g < G; g += gridDim.x) {
        const int64_t e0 = seg_ptr[g], e1 = seg_ptr[g + 1];
        const int lo = (int)hit_ptr[g], nh = (int)(hit_ptr[g + 1] - hit_ptr[g]);
        __syncthreads();
        for (int i = threadIdx.x; i < nh; i += B) {
            const int nw = inv[lo + i];
            const I2 p = ptr[nw];
            cin[i] = p.a;
            cout[i] = p.b;
            if (INV_LDS) linv[i] = nw;
        }
        __syncthreads();
        // (four segments per thread in flight; with the new ids in LDS the only global latency left is src / dst)
        for (int pass = 0; pass < 2; ++pass)
            for (int64_t jb = e0 + threadIdx.x; jb < e1; jb += 4 * (int64_t)B) {
                int sv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = jb + (int64_t)u * B;
                    sv[u] = j < e1 ? __builtin_nontemporal_load(src + j) : -1;
                    dv[u] = j < e1 ? __builtin_nontemporal_load(dst + j) : -1;
                }
                int sn[4], dn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = sv[u] >= 0;
                    sn[u] = ok ? (INV_LDS ? linv[sv[u] - lo] : inv[sv[u]]) : npad;
                    dn[u] = ok ? (INV_LDS ? linv[dv[u] - lo] : inv[dv[u]]) : npad;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = jb + (int64_t)u * B;
                    if (j >= e1) continue;
                    if (pass == 0) {
                        __builtin_nontemporal_store(sn[u], src_new + j);
                        __builtin_nontemporal_store(dn[u], dst_new + j);
                    }
                    if (sv[u] < 0) continue;
                    if (pass == 0) pin[atomicAdd(&cin[dv[u] - lo], 1)] = I2{(int)j, sn[u]};
                    else pout[atomicAdd(&cout[sv[u] - lo], 1)] = I2{(int)j, dn[u]};
                }
            }
    }
}

// 16 lanes per padded hit: every lane takes one (segment, other end) pair of the hit's list, counts the pairs with a
// smaller segment id (16 pairs at a time, passed round the group) and stores the other end at that rank.
__global__ __launch_bounds__(TB) void pb_sort_lists(const I2 *__restrict__ degn, const I2 *__restrict__ ptr,
                                                    const I2 *__restrict__ pin, const I2 *__restrict__ pout, int *sv_in,
                                                    int *sv_out, const int64_t *__restrict__ hdr)
{
    if (hdr[H_STATUS] || hdr[H_LISTMODE]) return;
    const int64_t n_pad = hdr[H_NPAD];
    const int l = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t h = grp; h < n_pad; h += ngrp) {
        const I2 dg = degn[h], p = ptr[h];
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const int deg = dir ? dg.b : dg.a, p0 = dir ? p.b : p.a;
            const I2 *__restrict__ pr = dir ? pout : pin;
            int *sv = dir ? sv_out : sv_in;
            for (int base = 0; base < deg; base += 16) {
                const bool have = base + l < deg;
                const I2 e = have ? pr[p0 + base + l] : I2{0x7FFFFFFF, 0};
                int rank = 0;
                for (int ob = 0; ob < deg; ob += 16) {
                    const int oj = ob == base ? e.a : (ob + l < deg ? pr[p0 + ob + l].a : 0x7FFFFFFF);
#pragma unroll
                    for (int k = 0; k < 16; ++k) rank += __shfl(oj, k, 16) < e.a ? 1 : 0;
                }
                if (have) sv[p0 + rank] = e.b;
            }
        }
    }
}

// ---- tile-local neighbour lists ---------------------------------------------------------------------------
// The scattered pairs of pb_graph_lists are what is left of a sort's cost.  They are not needed when the segments
// of a tile's lists lie together in the caller's order - and the reference emits segments per layer pair
// (gnn/graph.py:80-93): the incoming segments of the hits of one detector level ARE one contiguous block.  So:
//   pb_graph_renumber (a workgroup per graph, the graph's new ids in LDS): src_new / dst_new, and for every tile the
//       range of segment ids that end (start) at one of its hits - min / max by one atomic pair per wave when all 64
//       consecutive segments of a wave instruction go to one tile, which is the normal case;
//   pb_lists_mode (one workgroup): LISTMODE 1 when the ranges of all tiles together are at most 16 E segments long (a
//       level cut into several tiles is read once per tile and direction: 4 bytes per segment, coalesced) and no
//       tile's lists exceed the LDS staging; otherwise (shuffled segments, hubs) pb_graph_lists / pb_sort_lists
//       run as before - both sets of kernels are launched, the wrong one returns at once;
//   pb_tile_lists (a workgroup per tile): walks the tile's range, keeps the segments that end (start) in the tile,
//       drops (segment, other end) into an LDS image of the tile's lists (LDS cursors), sorts every list by segment
//       id (16 lanes per hit, ranks by counting) and streams the other ends out to their final places: the only
//       global traffic is src_new / dst_new once per direction and the lists once.
constexpr int kTileListEntries = 16384;               // staged (segment, other end) pairs per tile and direction: 128 KB
struct TR { int in_lo, in_nhi, out_lo, out_nhi; };    // (nhi = -max: one memset to 0x7F.. initialises all four for atomicMin)

template <bool INV_LDS>
__global__ __launch_bounds__(1024) void pb_graph_renumber(const int *__restrict__ src, const int *__restrict__ dst,
                                                          const int64_t *__restrict__ hit_ptr,
                                                          const int64_t *__restrict__ seg_ptr, int64_t G, int n, int cap,
                                                          const int *__restrict__ inv, const int *__restrict__ slice_tile,
                                                          int *src_new, int *dst_new, TR *trange, int64_t *hdr)
{
    if (hdr[H_STATUS]) return;
    extern __shared__ int lds[];
    int *linv = lds;
    // Small tiles (few graphs: plan.py cuts a batch into >= 512 tiles) put a level's hits into many tiles, and a wave's 64
    // consecutive segments then end in a dozen of them: a dozen rounds of the range loop below per instruction (1 ms
    // for four detector graphs) for ranges pb_lists_mode would refuse anyway.  A wave that meets more than sixteen tiles
    // raises H_SPARSE (pb_lists_mode: scattered pairs) and stops keeping ranges.
    bool ranges = true;
    const int B = (int)blockDim.x;
    const int npad = inv[n];
    for (int64_t g = blockIdx.x; g < G; g += gridDim.x) {
        const int64_t e0 = seg_ptr[g], e1 = seg_ptr[g + 1];
        const int lo = (int)hit_ptr[g], nh = (int)(hit_ptr[g + 1] - hit_ptr[g]);
        __syncthreads();
        if (INV_LDS)
            for (int i = threadIdx.x; i < nh; i += B) linv[i] = inv[lo + i];
        __syncthreads();
        const int64_t span = 4 * (int64_t)B;
        auto load4 = [&](int64_t jb, int *sv, int *dv) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = jb + 64 * u;
                sv[u] = j < e1 ? __builtin_nontemporal_load(src + j) : -1;
                dv[u] = j < e1 ? __builtin_nontemporal_load(dst + j) : -1;
            }
        };
        int sv[4], dv[4];
        load4(e0 + (threadIdx.x & ~63) * 4 + (threadIdx.x & 63), sv, dv);
        for (int64_t jb = e0 + (threadIdx.x & ~63) * 4 + (threadIdx.x & 63); jb - (threadIdx.x & 63) < e1;
             jb += span) {                        // a wave takes 4 x 64 consecutive segments; the next four in flight
            int sx[4], dx[4];
            load4(jb + span, sx, dx);
            int sn[4], dn[4], t_out[4], t_in[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = sv[u] >= 0;
                sn[u] = ok ? (INV_LDS ? linv[sv[u] - lo] : inv[sv[u]]) : npad;
                dn[u] = ok ? (INV_LDS ? linv[dv[u] - lo] : inv[dv[u]]) : npad;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {                 // (all eight tile look-ups in flight before anything waits)
                const bool ok = sv[u] >= 0;
                t_out[u] = ok ? slice_tile[sn[u] / SLICE] : -1;
                t_in[u] = ok ? slice_tile[dn[u] / SLICE] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = jb + 64 * u;
                if (j < e1) {
                    __builtin_nontemporal_store(sn[u], src_new + j);
                    __builtin_nontemporal_store(dn[u], dst_new + j);
                }
            }
            // one pair of atomics per DISTINCT tile among a wave instruction's 64 consecutive segments (one tile in the
            // normal case, two or three where a level is cut into several tiles; per-lane atomics on those few words
            // serialised at the L2: 4 ms for c3 x 32)
            const int lane = (int)(threadIdx.x & 63);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = (int)(jb + 64 * u);
                const bool ok = sv[u] >= 0;
                if (!ranges) continue;                                        // (wave-uniform)
#pragma unroll
                for (int dir = 0; dir < 2; ++dir) {
                    const int t = dir ? t_out[u] : t_in[u];
                    unsigned long long rem = __ballot(ok);
                    int trips = 0;
                    while (rem) {                                             // (wave-uniform)
                        if (++trips > 16) {
                            if (lane == 0) hdr[H_SPARSE] = 1;
                            ranges = false;
                            break;
                        }
                        const int leader = __ffsll((long long)rem) - 1;
                        const int tl = __shfl(t, leader, 64);
                        const unsigned long long same = __ballot(ok && t == tl);
                        const int jf = __shfl(jj, __ffsll((long long)same) - 1, 64);
                        const int jl = __shfl(jj, 63 - __clzll((long long)same), 64);
                        if (lane == leader) {
                            atomicMin(dir ? &trange[tl].out_lo : &trange[tl].in_lo, jf);
                            atomicMin(dir ? &trange[tl].out_nhi : &trange[tl].in_nhi, -jl);
                        }
                        rem &= ~same;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sv[u] = sx[u]; dv[u] = dx[u]; }
        }
    }
}

__global__ __launch_bounds__(1024) void pb_lists_mode(const TR *__restrict__ trange, const int *__restrict__ tpad_off,
                                                      const I2 *__restrict__ ptr, int64_t E, int force_scatter, int64_t *hdr)
{
    __shared__ int red[16];
    if (hdr[H_STATUS]) return;
    const int nt = (int)hdr[H_NTILES];
    long long len = 0;
    int big = 0;
    for (int t = threadIdx.x; t < nt; t += 1024) {
        const TR r = trange[t];
        if (r.in_lo <= -r.in_nhi) len += (long long)(-r.in_nhi) - r.in_lo + 1;
        if (r.out_lo <= -r.out_nhi) len += (long long)(-r.out_nhi) - r.out_lo + 1;
        const I2 a = ptr[tpad_off[t]], b = ptr[tpad_off[t + 1]];
        if (b.a - a.a > kTileListEntries || b.b - a.b > kTileListEntries) big = 1;
    }
    // (sum in units of 64 segments so that it fits the int reduction: E < 2^30)
    const int tot = block_reduce_i((int)((len + 63) >> 6), red, OpAdd());
    big = block_reduce_i(big, red, OpOr());
    if (threadIdx.x == 0)
        hdr[H_LISTMODE] = (!force_scatter && !big && !hdr[H_SPARSE] && (long long)tot * 64 <= 16 * (long long)E + 64ll * nt + 65536) ? 1 : 0;
}

__global__ __launch_bounds__(1024) void pb_tile_lists(const int *__restrict__ src_new, const int *__restrict__ dst_new,
                                                      const TR *__restrict__ trange, const int *__restrict__ tpad_off,
                                                      const I2 *__restrict__ degn, const I2 *__restrict__ ptr, int *sv_in,
                                                      int *sv_out, const int *__restrict__ sbase, int iter_records,
                                                      int *t_desc, int64_t *hdr)
{
    if (hdr[H_STATUS] || !hdr[H_LISTMODE]) return;
    __shared__ int wred[16][2];
    int m_rec = 0, m_in = 0, m_out = 0, n_lds = 0;      // (thread 0: this workgroup's tiles, as in pb_tile_windows)
    extern __shared__ int lds[];
    I2 *stage = reinterpret_cast<I2 *>(lds);                         // [kTileListEntries]
    int *cur = lds + 2 * kTileListEntries;                           // [kMaxSlicesPerTile * SLICE]
    int *start = cur + kMaxSlicesPerTile * SLICE;                    // [kMaxSlicesPerTile * SLICE]
    const int nt = (int)hdr[H_NTILES];
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4, ngrp = (int)(blockDim.x >> 4);
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int h0 = tpad_off[t], h1 = tpad_off[t + 1], nh = h1 - h0;
        const TR r = trange[t];
        int wlo[2] = {0x7FFFFFFF, 0x7FFFFFFF}, whi[2] = {-1, -1};    // the tile's windows (pb_tile_windows), thread 0
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const int j0 = dir ? r.out_lo : r.in_lo, j1 = dir ? -r.out_nhi : -r.in_nhi;     // inclusive
            if (j0 > j1) continue;                                   // (workgroup-uniform: no segment ends / starts here)
            const int *__restrict__ key = dir ? src_new : dst_new;
            const int *__restrict__ oth = dir ? dst_new : src_new;
            int *sv = dir ? sv_out : sv_in;
            const int P0 = dir ? ptr[h0].b : ptr[h0].a;
            __syncthreads();
            const int Lt = (dir ? ptr[h1].b : ptr[h1].a) - P0;         // entries of the tile's lists
            for (int i = threadIdx.x; i < nh; i += blockDim.x) {
                const int p = (dir ? ptr[h0 + i].b : ptr[h0 + i].a) - P0;
                cur[i] = p;
                start[i] = p;
            }
            __syncthreads();
            const bool dense = j1 - j0 < 2 * Lt;                       // most segments of the range belong here
            for (int jb = j0 + (int)threadIdx.x; jb <= j1; jb += 4 * (int)blockDim.x) {
                int kv[4], ov[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = jb + u * (int)blockDim.x;
                    kv[u] = j <= j1 ? key[j] : -1;
                    ov[u] = (dense && j <= j1) ? oth[j] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (kv[u] >= h0 && kv[u] < h1) {
                        const int j = jb + u * (int)blockDim.x;
                        stage[atomicAdd(&cur[kv[u] - h0], 1)] = I2{j, dense ? ov[u] : oth[j]};
                    }
            }
            __syncthreads();
            {   // the window of this direction: min / max new id over all other ends (they are all in `stage` now)
                int mn = 0x7FFFFFFF, mx = -1;
                for (int k = threadIdx.x; k < Lt; k += blockDim.x) {
                    const int v = stage[k].b;
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const int x = __shfl_xor(mn, o, 64), y = __shfl_xor(mx, o, 64);
                    mn = x < mn ? x : mn;
                    mx = y > mx ? y : mx;
                }
                if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = mn; wred[threadIdx.x >> 6][1] = mx; }
                __syncthreads();
                if (threadIdx.x == 0)
                    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
                        wlo[dir] = wred[w][0] < wlo[dir] ? wred[w][0] : wlo[dir];
                        whi[dir] = wred[w][1] > whi[dir] ? wred[w][1] : whi[dir];
                    }
            }
            for (int i = grp; i < nh; i += ngrp) {
                const int p0 = start[i], deg = cur[i] - p0;          // (the cursor stands behind the hit's list now)
                for (int base = 0; base < deg; base += 16) {
                    const bool have = base + l < deg;
                    const I2 e = have ? stage[p0 + base + l] : I2{0x7FFFFFFF, 0};
                    int rank = 0;
                    for (int ob = 0; ob < deg; ob += 16) {
                        const int oj = ob == base ? e.a : (ob + l < deg ? stage[p0 + ob + l].a : 0x7FFFFFFF);
#pragma unroll
                        for (int k = 0; k < 16; ++k) rank += __shfl(oj, k, 16) < e.a ? 1 : 0;
                    }
                    if (have) sv[P0 + p0 + rank] = e.b;
                }
            }
        }
        if (threadIdx.x == 0) {                                     // the tile's descriptor, as pb_tile_windows writes it
            const int icnt = whi[0] >= 0 ? whi[0] - wlo[0] + 1 : 0, ocnt = whi[1] >= 0 ? whi[1] - wlo[1] + 1 : 0;
            const int mode = (icnt + ocnt + 2) <= iter_records ? 1 : 0;
            int *d = t_desc + (int64_t)t * 8;
            d[0] = h0 / SLICE; d[1] = h1 / SLICE;
            d[2] = icnt > 0 ? wlo[0] : 0; d[3] = icnt;
            d[4] = ocnt > 0 ? wlo[1] : 0; d[5] = ocnt;
            d[6] = mode; d[7] = sbase[t];
            if (mode) {
                m_rec = icnt + ocnt + 2 > m_rec ? icnt + ocnt + 2 : m_rec;
                m_in = icnt > m_in ? icnt : m_in;
                m_out = ocnt > m_out ? ocnt : m_out;
                ++n_lds;
            }
        }
    }
    if (threadIdx.x == 0 && n_lds) {
        hdr_max(hdr, H_LDSREC, m_rec);
        hdr_max(hdr, H_LDSIN, m_in);
        hdr_max(hdr, H_LDSOUT, m_out);
        hdr_add(hdr, H_NLDSTILES, n_lds);
    }
}

__global__ __launch_bounds__(TB) void pb_fill_i32(int *a, int64_t n, int v) { GS_LOOP(i, n) a[i] = v; }

// (only hits without incoming segments use it: one segment in ten of a layered graph)
__global__ __launch_bounds__(TB) void pb_down(const int *__restrict__ src, const int *__restrict__ dst, int64_t E, int n,
                                              const int *__restrict__ level, const int *__restrict__ deg_in, int *down)
{
    GS_LOOP(j, E) {
        const int s = src[j], d = dst[j];
        if (seg_ok(s, d, n) && deg_in[s] == 0) atomicMin(&down[s], level[d]);
    }
}

// hits without incoming segments sit one level below their nearest end hit; sort key 1
__global__ __launch_bounds__(TB) void pb_key1(int64_t n, const int *__restrict__ deg_in, const int *__restrict__ deg_out,
                                              const int *__restrict__ gid, int *level, const int *__restrict__ down,
                                              unsigned long long *key, int *iota, int *hkey, int64_t *hdr)
{
    __shared__ int red[TB / 64];
    int bad = 0, lmax = 0;
    GS_LOOP(i, n) {
        const int di = deg_in[i], dout = deg_out[i];
        int lv = level[i];
        if (di == 0 && dout > 0) {
            lv = down[i] - 1;
            if (lv < 0) lv = 0;
            level[i] = lv;
        }
        if (di > 0xFFFF || dout > 0xFFFF || lv > 127) bad = 1;
        lmax = lv > lmax ? lv : lmax;
        key[i] = ((unsigned long long)(unsigned)gid[i] << 39) | ((unsigned long long)(lv & 127) << 32) |
                 ((unsigned long long)(0xFFFF - (di & 0xFFFF)) << 16) | (unsigned long long)(0xFFFF - (dout & 0xFFFF));
        iota[i] = (int)i;
        hkey[i] = gid[i] * 128 + (lv & 127);       // (graph, level) of a hit in one word: the chunk runs' key
    }
    bad = block_reduce_i(bad, red, OpOr());
    lmax = block_reduce_i(lmax, red, OpMax());
    if (threadIdx.x == 0 && bad) set_status(hdr, ST_DEGREE);
    if (threadIdx.x == 0) hdr_max(hdr, H_MAXLEVEL, lmax);
}

__global__ __launch_bounds__(TB) void pb_unit_flags(int64_t n, const unsigned long long *__restrict__ key, int *flag)
{
    GS_LOOP(p, n + 1) flag[p] = (p < n && (p == 0 || (key[p] >> 32) != (key[p - 1] >> 32))) ? 1 : 0;
}

// ordered compaction: out[scan[p]] = p where flag[p]; count = scan[n] (flag / scan have n + 1 entries)
__global__ __launch_bounds__(TB) void pb_compact(int64_t n, const int *__restrict__ flag, const int *__restrict__ scan,
                                                 int *out, int64_t *hdr, int slot)
{
    // a status set upstream (FAST_MISS: pb_graph_levels left a graph's keys / run flags unwritten) means flags and
    // scan may hold anything: nothing is compacted, the count is 0 and the kernels behind see empty lists
    if (hdr[H_STATUS]) { if (blockIdx.x == 0 && threadIdx.x == 0) hdr[slot] = 0; return; }
    GS_LOOP(p, n + 1) {
        if (p < n) {
            if (flag[p]) out[scan[p]] = (int)p;
        } else {
            out[scan[n]] = (int)n;               // end sentinel
            hdr[slot] = scan[n];
        }
    }
}

// plan.py's sequential packing of (graph, level) units into tiles of <= T hits.  The greedy rule
// (start a new tile when the next unit would not fit; a unit larger than T is split) makes a tile
// that starts at unit i end before the first unit j > i with ustart[j + 1] - ustart[i] > T: all
// threads compute that jump for the units of a chunk staged in LDS (binary search), one thread then
// follows the chain - one LDS read per TILE instead of a dependent step per unit.  A tile still open
// at the end of a chunk is carried over by its start position.
// (The walk itself is sequential - a tile starts where the previous one ended.  jmp[i] holds the next
// FOUR cut positions after a tile that starts at unit i, when all four steps are ordinary ones inside
// the chunk: the loop-carried chain of the one walking thread is then one 16-byte LDS read per four
// tiles instead of four dependent 4-byte reads per tile - 0.43 -> 0.1 ms for the 2816 units of c3 x 256.)
constexpr int kCutChunk = 2048;
// Up to kParUnits units (8192: 96 KB of LDS) the same cut without the sequential walk: the tile starts are the ORBIT
// of unit 0 under "next tile start" (nxt[i] as below; i + 1 after a unit that is split), and an orbit is marked by
// pointer doubling - after round r the first 2^r tile starts carry a mark and jump[i] = nxt^(2^r)(i): 13 rounds of
// two barriers instead of one dependent LDS read per tile by one thread (0.22 -> 0.03 ms for the 2816 units of c3 x
// 256).  A marked unit emits its start and, when it is larger than T, the cuts inside it; an exclusive scan of those
// counts places them.  Same bounds as the walk below, entry for entry (which stays for more units than fit).
constexpr int kParUnits = 8192;
__global__ __launch_bounds__(1024) void pb_cut_tiles_par(const int *__restrict__ ustart, int n, int T, int *tile_bounds,
                                                         int64_t *hdr, int nt_max)
{
    extern __shared__ int lds[];
    int *st = lds;                                 // [kParUnits + 1]
    int *jump = st + kParUnits + 1;                // [kParUnits]
    int *mark = jump + kParUnits;                  // [kParUnits]
    __shared__ int part[1024];
    const int nu = (int)hdr[H_NUNITS];
    if (nu > kParUnits) return;
    if (nu == 0) {
        if (threadIdx.x == 0) { tile_bounds[0] = 0; if (n) tile_bounds[1] = n; hdr[H_NTILES] = n ? 1 : 0; }
        return;
    }
    constexpr int EPT = kParUnits / 1024;
    for (int i = threadIdx.x; i <= nu; i += 1024) st[i] = ustart[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nu; i += 1024) {
        int nx = i + 1;
        if (st[i + 1] - st[i] <= T) {              // first j in (i, nu) with st[j + 1] > st[i] + T, or nu
            const int lim = st[i] + T;
            int lo = i, hi = nu;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (st[mid + 1] > lim) hi = mid; else lo = mid + 1;
            }
            nx = lo;
        }
        jump[i] = nx;
        mark[i] = i == 0 ? 1 : 0;
    }
    __syncthreads();
    for (int r = 0; (1 << r) < nu; ++r) {
        int q2[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = threadIdx.x + e * 1024;
            q2[e] = nu;
            if (i < nu) {
                const int q = jump[i];
                if (q < nu) {
                    if (mark[i]) mark[q] = 1;      // (a mark that is seen and passed on within the round is a tile start too)
                    q2[e] = jump[q];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = threadIdx.x + e * 1024;
            if (i < nu) jump[i] = q2[e];
        }
        __syncthreads();
    }
    // boundaries per tile start, scanned (thread t owns units [t EPT, (t + 1) EPT))
    int c[EPT], sum = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = (int)threadIdx.x * EPT + e;
        c[e] = 0;
        if (i < nu && mark[i]) {
            const int sz = st[i + 1] - st[i];
            c[e] = 1 + (sz > T ? (sz - 1) / T : 0);
        }
        sum += c[e];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {            // Hillis-Steele inclusive scan
        int x = 0;
        if ((int)threadIdx.x >= o) x = part[threadIdx.x - o];
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    const int total = part[1023];
    if (total > nt_max) {
        if (threadIdx.x == 0) { set_status(hdr, ST_TILES); hdr[H_NTILES] = 0; }
        return;
    }
    int at = part[threadIdx.x] - sum;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = (int)threadIdx.x * EPT + e;
        for (int q = 0; q < c[e]; ++q) tile_bounds[at + q] = st[i] + q * T;
        at += c[e];
    }
    if (threadIdx.x == 0) { tile_bounds[total] = n; hdr[H_NTILES] = total; }
}

__global__ __launch_bounds__(1024) void pb_cut_tiles(const int *__restrict__ ustart, int n, int T, int *tile_bounds,
                                                     int64_t *hdr, int nt_max)
{
    __shared__ int st[kCutChunk + 1], nxt[kCutChunk];
    __shared__ int4 jmp[kCutChunk];
    const int nu = (int)hdr[H_NUNITS];
    if (nu <= kParUnits) return;                   // (pb_cut_tiles_par cut them)
    int nb = 0, last = 0;
    bool over = false;
    auto push = [&](int v) {
        if (nb <= nt_max) tile_bounds[nb] = v; else over = true;
        ++nb;
        last = v;
    };
    // first j in [lo, cnt) with st[j + 1] > lim, or cnt
    auto first_over = [&](int lo, int cnt, int lim) {
        int hi = cnt;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (st[mid + 1] > lim) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    if (threadIdx.x == 0) push(0);
    int anchor = -1;                               // start position of a tile carried into this chunk
    for (int b0 = 0; b0 < nu; b0 += kCutChunk) {
        const int cnt = nu - b0 < kCutChunk ? nu - b0 : kCutChunk;
        __syncthreads();
        for (int i = threadIdx.x; i <= cnt; i += blockDim.x) st[i] = ustart[b0 + i];
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) nxt[i] = first_over(i, cnt, st[i] + T);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            int4 q = make_int4(-1, -1, -1, -1);
            auto step = [&](int u) { return (u >= 0 && u < cnt && st[u + 1] - st[u] <= T && nxt[u] < cnt) ? nxt[u] : -1; };
            q.x = step(i);
            q.y = step(q.x);
            q.z = step(q.y);
            q.w = step(q.z);
            jmp[i] = q;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int i = 0;
            if (anchor >= 0) {                     // the open tile ends inside this chunk, or goes on
                const int j = first_over(0, cnt, anchor + T);
                if (j < cnt) { push(st[j]); anchor = -1; }
                i = j;
            }
            while (i < cnt) {
                const int4 q = jmp[i];
                if (q.w >= 0) {                    // four ordinary steps at once
                    push(st[q.x]); push(st[q.y]); push(st[q.z]); push(st[q.w]);
                    i = q.w;
                    continue;
                }
                const int s0 = st[i], sz = st[i + 1] - s0;
                if (sz > T) {                      // split a big unit (a tile boundary stands at s0 already)
                    for (int a = s0 + T; a < s0 + sz; a += T) push(a);
                    push(s0 + sz);
                    ++i;
                    continue;
                }
                const int j = nxt[i];
                if (j < cnt) { push(st[j]); i = j; }
                else { anchor = s0; i = cnt; }     // still open at the end of the chunk
            }
        }
    }
    if (threadIdx.x == 0) {
        if (n && last != n) push(n);
        if (over) { set_status(hdr, ST_TILES); nb = 1; }
        hdr[H_NTILES] = nb - 1;
    }
}


// The second hit sort of the global form - (tile, -ceil(in / 4), -out), stable in the first sort's order - is local
// to a tile of at most kMaxSlicesPerTile * SLICE hits: one workgroup per tile sorts (key, position) words in LDS.
__global__ __launch_bounds__(TB) void pb_tile_sort(int64_t n, const int *__restrict__ base, const int *__restrict__ deg_in,
                                                   const int *__restrict__ deg_out, const int *__restrict__ tile_bounds,
                                                   const int64_t *__restrict__ hdr, int *tpos, int *oor)
{
    __shared__ unsigned long long a[kMaxSlicesPerTile * SLICE];
    if (hdr[H_STATUS]) return;
    const int nt = (int)hdr[H_NTILES];
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int p0 = tile_bounds[t], cnt = tile_bounds[t + 1] - p0;
        if (cnt <= 0 || cnt > kMaxSlicesPerTile * SLICE) continue;      // (pb_cut_tiles never makes one)
        const int M = pow2_ceil(cnt);
        __syncthreads();
        for (int i = threadIdx.x; i < M; i += TB) {
            unsigned long long k = ~0ull;
            if (i < cnt) {
                const int old = base[p0 + i];
                k = ((unsigned long long)(0xFFFF - ((deg_in[old] + 3) >> 2)) << 28) |
                    ((unsigned long long)(0xFFFF - deg_out[old]) << 12) | (unsigned long long)i;
                tpos[p0 + i] = t;
            }
            a[i] = k;
        }
        __syncthreads();
        bitonic_sort_lds(a, M);
        for (int r = threadIdx.x; r < cnt; r += TB) oor[p0 + r] = base[p0 + (int)(a[r] & 0xFFFull)];
    }
}

// per tile: padded size, padded offset (exclusive scan), slices, schedule base; one workgroup
__global__ __launch_bounds__(1024) void pb_tile_offsets(const int *__restrict__ tile_bounds, int *tpad_off, int *sbase,
                                                        int64_t *hdr, int64_t np_max)
{
    __shared__ int sa[1024], sb[1024];
    const int nt = (int)hdr[H_NTILES];
    int carry_a = 0, carry_b = 0, tmax = 0;
    for (int b0 = 0; b0 < nt; b0 += 1024) {
        const int t = b0 + threadIdx.x;
        int pad = 0, sch = 0;
        if (t < nt) {
            pad = (tile_bounds[t + 1] - tile_bounds[t] + SLICE - 1) / SLICE * SLICE;
            sch = ((pad / SLICE + 15) / 16) * 16;
            tmax = pad > tmax ? pad : tmax;
        }
        sa[threadIdx.x] = pad;
        sb[threadIdx.x] = sch;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {            // Hillis-Steele inclusive scan
            int xa = 0, xb = 0;
            if ((int)threadIdx.x >= o) { xa = sa[threadIdx.x - o]; xb = sb[threadIdx.x - o]; }
            __syncthreads();
            sa[threadIdx.x] += xa;
            sb[threadIdx.x] += xb;
            __syncthreads();
        }
        if (t < nt) {
            tpad_off[t] = carry_a + sa[threadIdx.x] - pad;
            sbase[t] = carry_b + sb[threadIdx.x] - sch;
        }
        carry_a += sa[1023];
        carry_b += sb[1023];
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(tmax, o, 64); tmax = x > tmax ? x : tmax; }
    if ((threadIdx.x & 63) == 0) hdr_max(hdr, H_TILEMAX, tmax);
    if (threadIdx.x == 0) {
        tpad_off[nt] = carry_a;
        sbase[nt] = carry_b;
        hdr[H_NPAD] = carry_a;
        hdr[H_NSCHED] = carry_b + 16;
        if (carry_a > np_max) set_status(hdr, ST_PAD);
    }
}

// rank r of the tile-major, degree-sorted order -> padded new id; degrees in new ids
__global__ __launch_bounds__(TB) void pb_new_ids(int64_t n, const int *__restrict__ oor, const int *__restrict__ tpos,
                                                 const int *__restrict__ tile_bounds, const int *__restrict__ tpad_off,
                                                 const int *__restrict__ deg_in, const int *__restrict__ deg_out,
                                                 int *inv, I2 *degn, int *slice_tile, const int64_t *__restrict__ hdr)
{
    if (hdr[H_STATUS]) return;
    GS_LOOP(r, n + 1) {
        if (r == n) { inv[n] = (int)hdr[H_NPAD]; continue; }
        const int old = oor[r], t = tpos[r];
        const int nw = tpad_off[t] + ((int)r - tile_bounds[t]);
        inv[old] = nw;
        degn[nw] = I2{deg_in[old], deg_out[old]};
        // tile of a slice: a tile is padded to whole slices, so every slice holds a real hit (its first one writes)
        if ((nw & (SLICE - 1)) == 0) slice_tile[nw / SLICE] = t;
    }
}

__global__ __launch_bounds__(TB) void pb_renumber(const int *__restrict__ src, const int *__restrict__ dst, int64_t E,
                                                  const int *__restrict__ inv, int n, int *src_new, int *dst_new,
                                                  const int64_t *__restrict__ hdr)
{
    if (hdr[H_STATUS]) return;
    GS_LOOP(j, E) {
        const int s = src[j];
        src_new[j] = inv[s >= 0 ? s : n];
        dst_new[j] = inv[s >= 0 ? dst[j] : n];
    }
}

// per slice: SELL-16 steps (max degree of its 16 hits), in entries: (16 L_in, 16 L_out, 16 ceil8 L_in, 16 ceil8 L_out)
__global__ __launch_bounds__(TB) void pb_slices(int64_t ns_max, const I2 *__restrict__ degn, I4 *sl4, int64_t *hdr)
{
    const int64_t ns = hdr[H_STATUS] ? 0 : hdr[H_NPAD] / SLICE;
    int mx = 0;
    GS_LOOP(s, ns_max + 1) {
        int li = 0, lo = 0;
        if (s < ns) {
#pragma unroll
            for (int i = 0; i < SLICE; ++i) {
                const I2 d = degn[s * SLICE + i];
                li = d.a > li ? d.a : li;
                lo = d.b > lo ? d.b : lo;
            }
        }
        sl4[s] = I4{li * SLICE, lo * SLICE, (li + 7) / 8 * 8 * SLICE, (lo + 7) / 8 * 8 * SLICE};
        mx = li > mx ? li : mx;
        mx = lo > mx ? lo : mx;
    }
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(mx, o, 64); mx = x > mx ? x : mx; }
    if ((threadIdx.x & 63) == 0 && mx) hdr_max(hdr, H_MAXSTEPS, mx);
}

// windows of a tile: min / max new id over the start hits of its incoming (end hits of its outgoing)
// segments - read from the sorted neighbour lists, no atomics.  One workgroup per tile.
__global__ __launch_bounds__(TB) void pb_tile_windows(const int *__restrict__ tpad_off, const int *__restrict__ sbase,
                                                      const I2 *__restrict__ degn, const I2 *__restrict__ ptr,
                                                      const int *__restrict__ sv_in, const int *__restrict__ sv_out,
                                                      int iter_records, int *t_desc, int64_t *hdr)
{
    if (hdr[H_STATUS] || hdr[H_LISTMODE]) return;      // (LISTMODE 1: pb_tile_lists wrote the descriptors)
    const int nt = (int)hdr[H_NTILES];
    __shared__ int red[4][TB / 64];
    int m_rec = 0, m_in = 0, m_out = 0, n_lds = 0;      // (thread 0: this workgroup's tiles; one set of atomics at the end)
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int h0 = tpad_off[t], h1 = tpad_off[t + 1];
        int ilo = 0x7FFFFFFF, ihi = -1, olo = 0x7FFFFFFF, ohi = -1;
        for (int h = h0 + threadIdx.x; h < h1; h += TB) {
            const I2 d = degn[h], p = ptr[h];
            for (int k = 0; k < d.a; ++k) { const int v = sv_in[p.a + k]; ilo = v < ilo ? v : ilo; ihi = v > ihi ? v : ihi; }
            for (int k = 0; k < d.b; ++k) { const int v = sv_out[p.b + k]; olo = v < olo ? v : olo; ohi = v > ohi ? v : ohi; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            int x = __shfl_xor(ilo, o, 64); ilo = x < ilo ? x : ilo;
            x = __shfl_xor(ihi, o, 64); ihi = x > ihi ? x : ihi;
            x = __shfl_xor(olo, o, 64); olo = x < olo ? x : olo;
            x = __shfl_xor(ohi, o, 64); ohi = x > ohi ? x : ohi;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            red[0][w] = ilo; red[1][w] = ihi; red[2][w] = olo; red[3][w] = ohi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < TB / 64; ++w) {
                ilo = red[0][w] < ilo ? red[0][w] : ilo; ihi = red[1][w] > ihi ? red[1][w] : ihi;
                olo = red[2][w] < olo ? red[2][w] : olo; ohi = red[3][w] > ohi ? red[3][w] : ohi;
            }
            const int icnt = ihi >= 0 ? ihi - ilo + 1 : 0, ocnt = ohi >= 0 ? ohi - olo + 1 : 0;
            const int mode = (icnt + ocnt + 2) <= iter_records ? 1 : 0;
            int *d = t_desc + (int64_t)t * 8;
            d[0] = h0 / SLICE; d[1] = h1 / SLICE;
            d[2] = icnt > 0 ? ilo : 0; d[3] = icnt;
            d[4] = ocnt > 0 ? olo : 0; d[5] = ocnt;
            d[6] = mode; d[7] = sbase[t];
            if (mode) {
                m_rec = icnt + ocnt + 2 > m_rec ? icnt + ocnt + 2 : m_rec;
                m_in = icnt > m_in ? icnt : m_in;
                m_out = ocnt > m_out ? ocnt : m_out;
                ++n_lds;
            }
        }
    }
    if (threadIdx.x == 0 && n_lds) {
        hdr_max(hdr, H_LDSREC, m_rec);
        hdr_max(hdr, H_LDSIN, m_in);
        hdr_max(hdr, H_LDSOUT, m_out);
        hdr_add(hdr, H_NLDSTILES, n_lds);
    }
}

// ---- edge chunks (plan._chunk_bounds) ---------------------------------------------------------------
// run starts of the (graph, start level) key; padded segments join the run before them: k[j] = key of
// the last valid segment at or before j (-1 when there is none).  The key of a hit is ONE word (hkey,
// written by pb_key1): one random 4-byte read per segment here, then a carry-forward scan ("the right
// operand unless it is -1"; keys are >= 0) and a compare of neighbours - where four gathers per
// segment (src[idx[j]], src[idx[j-1]], gid[], level[]) cost 0.46 ms at c3 x 256.
__global__ __launch_bounds__(TB) void pb_seg_keys(const int *__restrict__ src, int64_t E, int n, const int *__restrict__ hkey,
                                                  int *key)
{
    GS_LOOP(j, E) {
        const int s = src[j];
        key[j] = (unsigned)s < (unsigned)n ? hkey[s] : -1;              // (>= n: ST_ENDPOINT is set)
    }
}

__global__ __launch_bounds__(TB) void pb_run_flags(int64_t E, const int *__restrict__ k, int *flag)
{
    GS_LOOP(j, E + 1) {
        int f = 0;
        if (j < E) f = (j == 0) ? 1 : (k[j] != k[j - 1] ? 1 : 0);
        flag[j] = f;
    }
}

// marks = run starts that begin or follow a run of >= CH/4 segments (+ 0 and E)
__global__ __launch_bounds__(TB) void pb_mark_flags(const int *__restrict__ rb, int64_t E, int CH, int *flag,
                                                    const int64_t *__restrict__ hdr)
{
    const int64_t nr = hdr[H_NRUNS];
    const int thr = CH / 4 > 1 ? CH / 4 : 1;
    GS_LOOP(r, E + 1) {
        int f = 0;
        if (r < nr) {
            const bool big = rb[r + 1] - rb[r] >= thr;
            const bool big_prev = r > 0 && rb[r] - rb[r - 1] >= thr;
            f = (r == 0 || big || big_prev) ? 1 : 0;
        }
        flag[r] = f;
    }
}

// marks[scan[r]] = rb[r] for flagged runs; marks[n_marks] = E
__global__ __launch_bounds__(TB) void pb_marks(const int *__restrict__ rb, int64_t E, const int *__restrict__ flag,
                                               const int *__restrict__ scan, int *marks, int64_t m_max, int64_t *hdr)
{
    const int64_t nr = hdr[H_NRUNS];
    const int64_t nm = scan[nr];
    if (nm + 1 > m_max) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { set_status(hdr, ST_CHUNKS); hdr[H_NMARKS] = 0; }
        return;
    }
    GS_LOOP(r, nr + 1) {
        if (r < nr) {
            if (flag[r]) marks[scan[r]] = rb[r];
        } else {
            marks[nm] = (int)E;
            hdr[H_NMARKS] = nm;
        }
    }
}

// The marks of the graph-local form, one workgroup: runs in order, 1024 at a time, flagged ones compacted behind a
// running count (a ballot prefix per wave, wave totals through LDS) - instead of a flag array and a scan over E + 1
// entries for what are a few thousand runs.  More than kFastMaxRuns runs (shuffled segments): FAST_MISS.
constexpr int64_t kFastMaxRuns = 1 << 18;
__global__ __launch_bounds__(1024) void pb_marks_fast(const int *__restrict__ rb, int64_t E, int CH, int *marks, int64_t m_max,
                                                      int64_t *hdr)
{
    __shared__ int wtot[16];
    if (hdr[H_STATUS]) { if (threadIdx.x == 0) hdr[H_NMARKS] = 0; return; }
    const int64_t nr = hdr[H_NRUNS];
    if (nr > kFastMaxRuns) { if (threadIdx.x == 0) { set_status(hdr, ST_FAST_MISS); hdr[H_NMARKS] = 0; } return; }
    const int thr = CH / 4 > 1 ? CH / 4 : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t carry = 0;
    bool over = false;
    for (int64_t r0 = 0; r0 < nr; r0 += 1024) {
        const int64_t r = r0 + threadIdx.x;
        int f = 0, pos = 0;
        if (r < nr) {
            pos = rb[r];
            const bool big = rb[r + 1] - pos >= thr;
            const bool big_prev = r > 0 && pos - rb[r - 1] >= thr;
            f = (r == 0 || big || big_prev) ? 1 : 0;
        }
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wtot[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wv) woff += t; tot += t; }
        if (f) {
            const int64_t at = carry + woff + before;
            if (at < m_max) marks[at] = pos; else over = true;
        }
        carry += tot;
    }
    if (__syncthreads_or(over ? 1 : 0) || carry + 1 > m_max) {
        if (threadIdx.x == 0) { set_status(hdr, ST_CHUNKS); hdr[H_NMARKS] = 0; }
        return;
    }
    if (threadIdx.x == 0) { marks[carry] = (int)E; hdr[H_NMARKS] = carry; }
}

// one workgroup: parts per mark interval, their scan, the chunk bounds (equal parts of <= CH)
__global__ __launch_bounds__(1024) void pb_chunk_bounds(const int *__restrict__ marks, int CH, int *parts, int *pscan, int *cb,
                                                        int64_t c_max, int64_t *hdr)
{
    __shared__ int sa[1024];
    if (hdr[H_STATUS]) { if (threadIdx.x == 0) hdr[H_NCHUNKS] = 0; return; }
    const int nm = (int)hdr[H_NMARKS];
    int carry = 0;
    for (int b0 = 0; b0 < nm; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        int p = 0;
        if (i < nm) p = (marks[i + 1] - marks[i] + CH - 1) / CH;
        sa[threadIdx.x] = p;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int x = 0;
            if ((int)threadIdx.x >= o) x = sa[threadIdx.x - o];
            __syncthreads();
            sa[threadIdx.x] += x;
            __syncthreads();
        }
        if (i < nm) { parts[i] = p; pscan[i] = carry + sa[threadIdx.x] - p; }
        carry += sa[1023];
        __syncthreads();
    }
    const bool over = carry + 1 > c_max;
    if (threadIdx.x == 0) {
        hdr[H_NCHUNKS] = over ? 0 : carry;
        if (over) set_status(hdr, ST_CHUNKS);
        cb[0] = 0;
    }
    if (over) return;
    __syncthreads();
    for (int i = threadIdx.x; i < nm; i += 1024) {
        const long long a = marks[i], len = marks[i + 1] - marks[i];
        const int p = parts[i], o = pscan[i];
        for (int k = 1; k <= p; ++k) cb[o + k] = (int)(a + ((long long)k * len) / p);
    }
}

// windows of a chunk of the final edge pass; one workgroup per chunk
__global__ __launch_bounds__(TB) void pb_chunk_windows(const int *__restrict__ cb, const int *__restrict__ src,
                                                       const int *__restrict__ src_new, const int *__restrict__ dst_new,
                                                       int edge_records, int *c_desc, int64_t *hdr)
{
    if (hdr[H_STATUS]) return;
    const int nc = (int)hdr[H_NCHUNKS];
    const int npad = (int)hdr[H_NPAD];
    __shared__ int red[4][TB / 64];
    int m_rows = 0, n_lds = 0;                          // (thread 0: this workgroup's chunks)
    for (int c = blockIdx.x; c < nc; c += gridDim.x) {
        const int e0 = cb[c], e1 = cb[c + 1];
        int slo = 0x7FFFFFFF, shi = -1, dlo = 0x7FFFFFFF, dhi = -1;
        // (a padded segment carries the id behind the last padded hit in both ends: no look at src; four steps in flight)
#pragma unroll 4
        for (int j = e0 + threadIdx.x; j < e1; j += TB) {
            const int s = src_new[j], d = dst_new[j];
            if (s < npad) {
                slo = s < slo ? s : slo; shi = s > shi ? s : shi;
                dlo = d < dlo ? d : dlo; dhi = d > dhi ? d : dhi;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            int x = __shfl_xor(slo, o, 64); slo = x < slo ? x : slo;
            x = __shfl_xor(shi, o, 64); shi = x > shi ? x : shi;
            x = __shfl_xor(dlo, o, 64); dlo = x < dlo ? x : dlo;
            x = __shfl_xor(dhi, o, 64); dhi = x > dhi ? x : dhi;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            red[0][w] = slo; red[1][w] = shi; red[2][w] = dlo; red[3][w] = dhi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < TB / 64; ++w) {
                slo = red[0][w] < slo ? red[0][w] : slo; shi = red[1][w] > shi ? red[1][w] : shi;
                dlo = red[2][w] < dlo ? red[2][w] : dlo; dhi = red[3][w] > dhi ? red[3][w] : dhi;
            }
            const int scnt = shi >= 0 ? shi - slo + 1 : 0, dcnt = dhi >= 0 ? dhi - dlo + 1 : 0;
            const int mode = (scnt + dcnt + 2) <= edge_records ? 1 : 0;
            int *d = c_desc + (int64_t)c * 8;
            d[0] = e0; d[1] = e1;
            d[2] = scnt > 0 ? slo : 0; d[3] = scnt;
            d[4] = dcnt > 0 ? dlo : 0; d[5] = dcnt;
            d[6] = mode; d[7] = 0;
            if (mode) {
                m_rows = scnt + dcnt + 2 > m_rows ? scnt + dcnt + 2 : m_rows;
                ++n_lds;
            }
        }
    }
    if (threadIdx.x == 0 && n_lds) {
        hdr_max(hdr, H_EDGEROWS, m_rows);
        hdr_add(hdr, H_NLDSCHUNKS, n_lds);
    }
}

__global__ void pb_sizes(const int64_t *__restrict__ hdr, const I4 *__restrict__ off4, gnn_plan_sizes_t *out, int64_t np_max,
                         int64_t ns_max)
{
    if (threadIdx.x || blockIdx.x) return;
    gnn_plan_sizes_t s;
    const int64_t st = hdr[H_STATUS];
    const int64_t npad = st ? 0 : hdr[H_NPAD];
    const int64_t ns = npad / SLICE;
    const I4 tot = off4[ns <= ns_max ? ns : 0];
    s.n_pad = npad; s.n_tiles = st ? 0 : hdr[H_NTILES]; s.n_slices = ns; s.n_chunks = hdr[H_NCHUNKS];
    s.in_total = tot.a; s.out_total = tot.b; s.in16_words = tot.c / 2; s.out16_words = tot.d / 2;
    s.n_sched = hdr[H_NSCHED];
    s.iter_lds_records = hdr[H_LDSREC]; s.edge_lds_rows = hdr[H_EDGEROWS];
    s.n_lds_tiles = hdr[H_NLDSTILES]; s.n_lds_chunks = hdr[H_NLDSCHUNKS];
    s.iter_lds_in = hdr[H_LDSIN]; s.iter_lds_out = hdr[H_LDSOUT];
    s.tile_hits_max = hdr[H_TILEMAX]; s.max_list_steps = hdr[H_MAXSTEPS];
    s.n_valid = hdr[H_NVALID]; s.max_level = hdr[H_MAXLEVEL];
    s.status = st;
    s.list_mode = hdr[H_LISTMODE];
    *out = s;
}

// ---- stage 2 kernels ------------------------------------------------------------------------------
template <int F_MAX>
__global__ __launch_bounds__(TB) void pb_fill_hits(int64_t n, int F, const float *__restrict__ X, const int *__restrict__ oor,
                                                   const int *__restrict__ inv, float *Xp, int *perm, unsigned *absmax)
{
    float mx[F_MAX];
#pragma unroll
    for (int k = 0; k < F_MAX; ++k) mx[k] = 0.0f;
    GS_LOOP(r, n) {
        const int old = oor[r], nw = inv[old];
        perm[nw] = old;
#pragma unroll
        for (int k = 0; k < F_MAX; ++k)
            if (k < F) {
                const float v = X[(int64_t)old * F + k];
                Xp[(int64_t)nw * F + k] = v;
                mx[k] = fmaxf(mx[k], fabsf(v));
            }
    }
    __shared__ int red[TB / 64];
#pragma unroll
    for (int k = 0; k < F_MAX; ++k) {
        if (k >= F) break;
        // >= 0: bit order = value order; one atomic per workgroup and feature
        const int v = block_reduce_i((int)__float_as_uint(mx[k]), red, OpMax());
        if (threadIdx.x == 0 && v > 0) atomicMax(&absmax[k], (unsigned)v);
    }
}

__global__ __launch_bounds__(TB) void pb_fill_offsets(int64_t ns, const I4 *__restrict__ off4, int *in_off, int *out_off,
                                                      int *in_off16, int *out_off16)
{
    GS_LOOP(s, ns + 1) {
        const I4 o = off4[s];
        in_off[s] = o.a; out_off[s] = o.b; in_off16[s] = o.c / 2; out_off16[s] = o.d / 2;
    }
}

// SELL-16 lists of one direction, 32-bit and 16-bit packed: one lane per padded hit, entry k of hit
// i of slice s at off[s] + 16 k + i (16 lanes write 64 contiguous bytes per step)
template <bool IN>
__global__ __launch_bounds__(TB) void pb_fill_lists(int64_t n_pad, const I2 *__restrict__ degn, const I2 *__restrict__ ptr,
                                                    const I4 *__restrict__ sl4, const I4 *__restrict__ off4,
                                                    const int *__restrict__ sv, const int *__restrict__ t_desc,
                                                    const int *__restrict__ slice_tile, int *nbr, int *nbr16)
{
    GS_LOOP(h, n_pad) {
        const int64_t s = h / SLICE;
        const int i = (int)(h % SLICE);
        const int t = slice_tile[s];
        const int *td = t_desc + (int64_t)t * 8;
        const bool lds = td[6] != 0;
        const int lo = IN ? td[2] : td[4], cnt = IN ? td[3] : td[5];
        const int null = lds ? cnt : (int)n_pad;
        const int deg = IN ? degn[h].a : degn[h].b;
        const int p0 = IN ? ptr[h].a : ptr[h].b;
        const I4 sl = sl4[s], of = off4[s];
        const int steps = (IN ? sl.a : sl.b) / SLICE, steps8 = (IN ? sl.c : sl.d) / SLICE;
        int *o32 = nbr + (IN ? of.a : of.b) + i;
        int *o16 = nbr16 + (IN ? of.c : of.d) / 2 + i;
        for (int k = 0; k < steps8; k += 8) {            // (steps8 is a multiple of 8: eight list entries in flight)
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k + u < deg ? sv[p0 + k + u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + u;
                const int x = kk < deg ? (lds ? v[u] - lo : v[u]) : null;
                v[u] = x;
                if (kk < steps) o32[kk * SLICE] = x;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2)
                o16[((k + u) >> 1) * SLICE] = (int)(((unsigned)v[u] & 0xFFFFu) | (((unsigned)v[u + 1] & 0xFFFFu) << 16));
        }
    }
}

__global__ __launch_bounds__(TB) void pb_copy_i32(const int *__restrict__ a, int *b, int64_t n) { GS_LOOP(i, n) b[i] = a[i]; }

// wave schedules of one tile (plan.py schedule / schedule_two_phase); one wavefront per tile.
// float64 arithmetic exactly as numpy evaluates it: no contraction into fma.
#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void pb_schedule(int nt, const int *__restrict__ t_desc, const I4 *__restrict__ sl4,
                                                  int *sched_a, int *sched_b)
{
    __shared__ int ga[kMaxSlicesPerTile], gb[kMaxSlicesPerTile], by_ab[kMaxSlicesPerTile], by_tot[kMaxSlicesPerTile];
    __shared__ int rk[kMaxSlicesPerTile];
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int *td = t_desc + (int64_t)t * 8;
        const int s0 = td[0], nsl = td[1] - td[0], sb = td[7];
        __syncthreads();
        for (int i = lane; i < nsl; i += 64) {
            const I4 sl = sl4[s0 + i];
            ga[i] = (sl.a / SLICE + 3) / 4;
            gb[i] = (sl.b / SLICE + 3) / 4;
        }
        __syncthreads();
        // sched_b: heaviest out-phase first (stable), dealt in snake order over the 16 waves
        for (int i = lane; i < nsl; i += 64) {
            int r = 0;
            for (int j = 0; j < nsl; ++j) r += (gb[j] > gb[i] || (gb[j] == gb[i] && j < i)) ? 1 : 0;
            const int rd = r / 16, c = r % 16;
            sched_b[sb + rd * 16 + ((rd & 1) ? 15 - c : c)] = s0 + i;
            // rank by (in groups, out groups) descending, stable
            int q = 0;
            for (int j = 0; j < nsl; ++j) {
                const bool before = ga[j] > ga[i] || (ga[j] == ga[i] && (gb[j] > gb[i] || (gb[j] == gb[i] && j < i)));
                q += before ? 1 : 0;
            }
            by_ab[q] = i;
            rk[i] = q;
        }
        __syncthreads();
        // inside each round of 16: heaviest total first, stable in by_ab order
        for (int i = lane; i < nsl; i += 64) {
            const int q = rk[i], rd = q / 16;
            const double tot = (double)ga[i] + ((double)gb[i] + 2.6);
            int r2 = 0;
            const int q0 = rd * 16, q1 = q0 + 16 < nsl ? q0 + 16 : nsl;
            for (int qq = q0; qq < q1; ++qq) {
                const int j = by_ab[qq];
                const double tj = (double)ga[j] + ((double)gb[j] + 2.6);
                r2 += (tj > tot || (tj == tot && qq < q)) ? 1 : 0;
            }
            by_tot[q0 + r2] = i;
        }
        __syncthreads();
        // greedy: the next slice goes to the free wave that raises max A + max B least
        double A = 0.0, B = 0.0;                    // lane w < 16 holds wave w's phase loads
        const double inf = __builtin_huge_val();
        const int rounds = (nsl + 15) / 16;
        for (int rd = 0; rd < rounds; ++rd) {
            bool used = false;
            for (int j = 0; j < 16; ++j) {
                const int idx = rd * 16 + j;
                if (idx >= nsl) break;
                const int sl = by_tot[idx];
                const double a = (double)ga[sl], b = (double)gb[sl] + 2.6;
                double mA = lane < 16 ? A : -inf, mB = lane < 16 ? B : -inf;
                for (int o = 8; o > 0; o >>= 1) {
                    const double xa = __shfl_xor(mA, o, 64), xb = __shfl_xor(mB, o, 64);
                    mA = xa > mA ? xa : mA;
                    mB = xb > mB ? xb : mB;
                }
                const double t1 = (A + a) > mA ? (A + a) : mA;
                const double t2 = (B + b) > mB ? (B + b) : mB;
                const double t3 = 1e-3 * (A + B);
                double sc = (t1 + t2) + t3;
                if (used || lane >= 16) sc = inf;
                int w = lane;
                for (int o = 8; o > 0; o >>= 1) {
                    const double xs = __shfl_xor(sc, o, 64);
                    const int xw = __shfl_xor(w, o, 64);
                    if (xs < sc || (xs == sc && xw < w)) { sc = xs; w = xw; }
                }
                w = __shfl(w, 0, 64);
                if (lane == w) { A = A + a; B = B + b; used = true; }
                if (lane == 0) sched_a[sb + rd * 16 + w] = s0 + sl;
            }
        }
    }
}
#pragma clang fp contract(on)

// endpoints of the final edge pass in the caller's segment order (window-relative in LDS chunks)
__global__ __launch_bounds__(TB) void pb_fill_chunks(int nc, const int *__restrict__ c_desc, const int *__restrict__ src,
                                                     const int *__restrict__ src_new, const int *__restrict__ dst_new,
                                                     int *chunks, int *src_st, int *dst_st, int *sd16)
{
    for (int c = blockIdx.x; c < nc; c += gridDim.x) {
        const int *d = c_desc + (int64_t)c * 8;
        if (threadIdx.x < 8) chunks[(int64_t)c * 8 + threadIdx.x] = d[threadIdx.x];
        const int e0 = d[0], e1 = d[1], slo = d[2], scnt = d[3], dlo = d[4], dcnt = d[5], mode = d[6];
        for (int j = e0 + threadIdx.x; j < e1; j += TB) {
            const bool ok = src[j] >= 0;
            int s = src_new[j], t = dst_new[j];
            unsigned w = 0;
            if (mode) {
                s = ok ? s - slo : scnt;
                t = ok ? t - dlo : dcnt;
                w = ((unsigned)t << 16) | ((unsigned)s & 0xFFFFu);
            }
            src_st[j] = s;
            dst_st[j] = t;
            sd16[j] = (int)w;
        }
    }
}

inline unsigned bit_width(uint64_t v)
{
    unsigned b = 0;
    while (v) { ++b; v >>= 1; }
    return b ? b : 1;
}

#define HIP_OK(CALL, WHAT)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (CALL);                                                                       \
        if (e_ != hipSuccess) return fail(-(int)e_, "%s failed: %s", WHAT, hipGetErrorString(e_));   \
    } while (0)

}  // namespace

extern "C" {

size_t gnn_plan_build_workspace_bytes(int64_t n_hits, int64_t n_segments, int32_t chunk_segments)
{
    if (n_hits <= 0 || n_segments <= 0 || chunk_segments <= 0) return 0;
    return carve(nullptr, n_hits, n_segments, chunk_segments).bytes + 256;
}

// seg_ptr != nullptr: the graph-local form (pb_graph_levels, pb_graph_lists); its FAST_MISS status sends the caller
// back here with seg_ptr = nullptr
static int plan_sizes_impl(const int32_t *src, const int32_t *dst, const int64_t *hit_ptr, const int64_t *seg_ptr,
                           int64_t max_graph_hits, int64_t max_graph_segments, int64_t n_hits,
                           int64_t n_segments, int64_t n_graphs, int32_t tile_hits, int32_t iter_records,
                           int32_t chunk_segments, int32_t edge_records, void *workspace, size_t workspace_bytes,
                           gnn_plan_sizes_t *sizes_out, void *stream)
{
    const int64_t n = n_hits, E = n_segments, G = n_graphs;
    if (!src || !dst || !hit_ptr || !workspace || !sizes_out || n <= 0 || E <= 0 || G <= 0 || tile_hits < SLICE ||
        tile_hits > kMaxSlicesPerTile * SLICE || chunk_segments <= 0 || iter_records < 0 || edge_records < 0)
        return fail(GNN_ERR_BADARG, "gnn_plan_build_sizes: bad argument");
    const bool fast = seg_ptr != nullptr;
    if (fast && (max_graph_hits <= 0 || max_graph_hits > kGraphCapHits || max_graph_segments < 0))
        return fail(GNN_ERR_UNSUPPORTED, "gnn_plan_build_sizes_graphs: a graph of %lld hits does not fit the graph-local form "
                    "(<= %d)", (long long)max_graph_hits, kGraphCapHits);
    if (n >= (1ll << 30) || E >= (1ll << 30) || G >= (1ll << 24))
        return fail(GNN_ERR_UNSUPPORTED, "gnn_plan_build_sizes: batch too large for 32-bit plan indices");
    if (workspace_bytes < gnn_plan_build_workspace_bytes(n, E, chunk_segments))
        return fail(GNN_ERR_WORKSPACE, "gnn_plan_build_sizes: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *wb = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const Ws w = carve(wb, n, E, chunk_segments);
    const Bounds b = bounds_of(n, E, chunk_segments);
    int *level = w.lvA;
    // the cap of the LDS tables: the largest graph, rounded (a fresh opt-in per size would cost more than it saves)
    const int cap = (int)((max_graph_hits + 1023) / 1024 * 1024 < kGraphCapHits ? (max_graph_hits + 1023) / 1024 * 1024
                                                                                : kGraphCapHits);
    const unsigned gthreads = max_graph_segments > 8192 ? 1024u : 256u;
    const unsigned ggrid = (unsigned)(G < 16384 ? G : 16384);
    if (fast) {
        static DevOnce attr_done;     // dynamic LDS above 64 KB must be opted into, once per device
        if (attr_done.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_graph_levels),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kGraphLdsBytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_graph_lists<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kGraphLdsBytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_graph_lists<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kGraphLdsBytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_graph_renumber<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kGraphLdsBytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_tile_lists),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kGraphLdsBytes);
        }
        HIP_OK(hipMemsetAsync(w.hdr, 0, (size_t)(reinterpret_cast<char *>(w.deg_in) - reinterpret_cast<char *>(w.hdr)), s),
               "memset");
        HIP_OK(hipMemsetAsync(w.degn, 0, (size_t)(b.np_max + 1) * sizeof(I2), s), "memset");
        GNN_LAUNCH_SH("pb_graph_levels", pb_graph_levels, ggrid, gthreads,
                      cap <= kGraphCapHits64 ? (size_t)pow2_ceil(cap) * 8 : (size_t)cap * 8, s, src, dst, hit_ptr, seg_ptr, G, n,
                      E, cap, w.deg_in, w.deg_out, w.gid, level, w.k64a, w.iota, w.hkey, w.mscan, w.base, w.uflag, w.hdr);
    } else {
    // header, sweep flags, degrees and both level buffers start at zero (adjacent in the workspace)
    HIP_OK(hipMemsetAsync(w.hdr, 0, (size_t)(reinterpret_cast<char *>(w.gid) - reinterpret_cast<char *>(w.hdr)), s), "memset");
    HIP_OK(hipMemsetAsync(w.degn, 0, (size_t)(b.np_max + 1) * sizeof(I2), s), "memset");
    {
        const int64_t nb = (E + kDegSegs - 1) / kDegSegs;
        GNN_LAUNCH("pb_degrees", pb_degrees, (unsigned)(nb < 2048 ? nb : 2048), 1024, s, src, dst, E, (int)n, w.deg_in,
                   w.deg_out, w.hdr);
    }
    GNN_LAUNCH("pb_gid", pb_gid, gs(n), TB, s, hit_ptr, G, n, w.gid);
    if (E <= kSmallSweepSegments)
        GNN_LAUNCH("pb_levels_small", pb_levels_small, 1, 1024, s, src, dst, (int)E, (int)n, level);
    else
        // in rounds of 12 sweeps (a 10-layer detector graph is done after 11); one 4-byte read-back per
        // round costs ~15 us, the 52 launches it usually saves cost 0.3 ms
    {
        unsigned char *lv8 = reinterpret_cast<unsigned char *>(w.lvB);   // (zeroed with the header)
        // the blocks' "dead" bytes live in the zeroed rest of lvB (4 n bytes, n of them levels)
        const int blk = E >= (4ll << 20) ? 4096 : 1024;
        const int64_t nblk = (E + blk - 1) / blk;
        const int64_t dead_off = (n + 255) / 256 * 256;
        unsigned char *dead = dead_off + nblk <= 4 * n ? lv8 + dead_off : nullptr;
        const unsigned sg = (unsigned)(nblk < 1 ? 1 : nblk > 8192 ? 8192 : nblk);
        for (int t = 1; t <= kMaxLevelIters; ++t) {
            GNN_LAUNCH("pb_level_sweep", pb_level_sweep, sg, TB, s, src, dst, E, (int)n, lv8, w.chg, t, dead, blk);
            if (t % kSweepRound == 0 && t < kMaxLevelIters) {
                int raised = 1;
                HIP_OK(hipMemcpyAsync(&raised, w.chg + t, sizeof(int), hipMemcpyDeviceToHost, s), "sweep flag read-back");
                HIP_OK(hipStreamSynchronize(s), "sweep flag read-back");
                if (!raised) break;
            }
        }
        GNN_LAUNCH("pb_widen", pb_widen, gs(n), TB, s, lv8, level, n);
    }
    GNN_LAUNCH("pb_fill_i32", pb_fill_i32, gs(n), TB, s, w.down, n, 0x7FFFFFFF);
    GNN_LAUNCH("pb_down", pb_down, gs(E), TB, s, src, dst, E, (int)n, level, w.deg_in, w.down);
    GNN_LAUNCH("pb_key1", pb_key1, gs(n), TB, s, n, w.deg_in, w.deg_out, w.gid, level, w.down, w.k64a, w.iota, w.hkey, w.hdr);
    }
    size_t tb = w.temp_bytes;
    if (!fast) {                          // (graph-local form: pb_graph_levels sorted each graph's hits in LDS)
        HIP_OK(rocprim::radix_sort_pairs(w.temp, tb, (const unsigned long long *)w.k64a, w.k64b, (const int *)w.iota, w.base,
                                         (size_t)n, 0u, 39u + bit_width((uint64_t)G), s, false), "hit sort 1");
        GNN_LAUNCH("pb_unit_flags", pb_unit_flags, gs(n + 1), TB, s, n, w.k64b, w.uflag);
    }
    tb = w.temp_bytes;
    HIP_OK(rocprim::exclusive_scan(w.temp, tb, (const int *)w.uflag, w.uscan, 0, (size_t)n + 1, rocprim::plus<int>(), s, false),
           "unit scan");
    GNN_LAUNCH("pb_compact", pb_compact, gs(n + 1), TB, s, n, w.uflag, w.uscan, w.ustart, w.hdr, (int)H_NUNITS);
    {
        static DevOnce cut_attr;
        if (cut_attr.need())
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_cut_tiles_par),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kGraphLdsBytes);
    }
    GNN_LAUNCH_SH("pb_cut_tiles_par", pb_cut_tiles_par, 1, 1024, (size_t)(3 * kParUnits + 1) * 4, s, w.ustart, (int)n,
                  (int)tile_hits, w.tile_bounds, w.hdr, (int)b.nt_max);
    GNN_LAUNCH("pb_cut_tiles", pb_cut_tiles, 1, 1024, s, w.ustart, (int)n, (int)tile_hits, w.tile_bounds, w.hdr,
               (int)b.nt_max);
    // (both forms: the second hit sort is local to a tile - one workgroup sorts it in LDS; the global form ran a
    // device-wide radix sort of (tile, -ceil(in / 4), -out) keys here until late round 3)
    GNN_LAUNCH("pb_tile_sort", pb_tile_sort, 4096, TB, s, n, w.base, w.deg_in, w.deg_out, w.tile_bounds, w.hdr, w.tpos, w.oor);
    GNN_LAUNCH("pb_tile_offsets", pb_tile_offsets, 1, 1024, s, w.tile_bounds, w.tpad_off, w.sbase, w.hdr, b.np_max);
    GNN_LAUNCH("pb_new_ids", pb_new_ids, gs(n + 1), TB, s, n, w.oor, w.tpos, w.tile_bounds, w.tpad_off, w.deg_in, w.deg_out,
               w.inv, w.degn, w.slice_tile, w.hdr);
    tb = w.temp_bytes;
    HIP_OK(rocprim::exclusive_scan(w.temp, tb, (const I2 *)w.degn, w.ptr, I2{0, 0}, (size_t)b.np_max + 1, PlusI2(), s, false),
           "degree scan");
    if (fast) {
        HIP_OK(hipMemsetAsync(w.trange, 0x7F, (size_t)(b.nt_max + 1) * sizeof(TR), s), "memset");
        if ((size_t)cap * 4 <= (size_t)kGraphLdsBytes / 2)
            GNN_LAUNCH_SH("pb_graph_renumber", pb_graph_renumber<true>, ggrid, gthreads, (size_t)cap * 4, s, src, dst, hit_ptr,
                          seg_ptr, G, (int)n, cap, w.inv, w.slice_tile, w.src_new, w.dst_new, reinterpret_cast<TR *>(w.trange),
                          w.hdr);
        else
            GNN_LAUNCH_SH("pb_graph_renumber", pb_graph_renumber<false>, ggrid, gthreads, 0, s, src, dst, hit_ptr, seg_ptr, G,
                          (int)n, cap, w.inv, w.slice_tile, w.src_new, w.dst_new, reinterpret_cast<TR *>(w.trange), w.hdr);
        const char *fs = getenv("GNN_PLAN_SCATTER_LISTS");       // (tests and A / B timing: the scattered-pairs form)
        GNN_LAUNCH("pb_lists_mode", pb_lists_mode, 1, 1024, s, reinterpret_cast<const TR *>(w.trange), w.tpad_off, w.ptr, E,
                   (fs && fs[0] == '1') ? 1 : 0, w.hdr);
        GNN_LAUNCH_SH("pb_tile_lists", pb_tile_lists, 4096, 1024, (size_t)(2 * kTileListEntries + 2 * kMaxSlicesPerTile * SLICE) * 4, s,
                      w.src_new, w.dst_new, reinterpret_cast<const TR *>(w.trange), w.tpad_off, w.degn, w.ptr, w.sv_in, w.sv_out,
                      w.sbase, (int)iter_records, w.t_desc, w.hdr);
        if ((size_t)cap * 12 <= (size_t)kGraphLdsBytes)
            GNN_LAUNCH_SH("pb_graph_lists", pb_graph_lists<true>, ggrid, gthreads, (size_t)cap * 12, s, src, dst, hit_ptr, seg_ptr,
                          G, (int)n, cap, w.inv, w.ptr, w.src_new, w.dst_new, w.pin, w.pout, w.hdr);
        else
            GNN_LAUNCH_SH("pb_graph_lists", pb_graph_lists<false>, ggrid, gthreads, (size_t)cap * 8, s, src, dst, hit_ptr, seg_ptr,
                          G, (int)n, cap, w.inv, w.ptr, w.src_new, w.dst_new, w.pin, w.pout, w.hdr);
        GNN_LAUNCH("pb_sort_lists", pb_sort_lists, 4096, TB, s, w.degn, w.ptr, w.pin, w.pout, w.sv_in, w.sv_out, w.hdr);
    } else {
        GNN_LAUNCH("pb_renumber", pb_renumber, gs(E), TB, s, src, dst, E, w.inv, (int)n, w.src_new, w.dst_new, w.hdr);
        const unsigned idbits = bit_width((uint64_t)b.np_max);
        tb = w.temp_bytes;
        HIP_OK(rocprim::radix_sort_pairs(w.temp, tb, (const int *)w.dst_new, w.kscr, (const int *)w.src_new, w.sv_in, (size_t)E,
                                         0u, idbits, s, false), "segment sort (in)");
        tb = w.temp_bytes;
        HIP_OK(rocprim::radix_sort_pairs(w.temp, tb, (const int *)w.src_new, w.kscr, (const int *)w.dst_new, w.sv_out, (size_t)E,
                                         0u, idbits, s, false), "segment sort (out)");
    }
    GNN_LAUNCH("pb_slices", pb_slices, gs(b.ns_max + 1), TB, s, b.ns_max, w.degn, w.sl4, w.hdr);
    tb = w.temp_bytes;
    HIP_OK(rocprim::exclusive_scan(w.temp, tb, (const I4 *)w.sl4, w.off4, I4{0, 0, 0, 0}, (size_t)b.ns_max + 1, PlusI4(), s, false),
           "slice scan");
    GNN_LAUNCH("pb_tile_windows", pb_tile_windows, 2048, TB, s, w.tpad_off, w.sbase, w.degn, w.ptr, w.sv_in, w.sv_out,
               (int)iter_records, w.t_desc, w.hdr);
    // edge chunks
    if (!fast) {
        GNN_LAUNCH("pb_seg_keys", pb_seg_keys, gs(E), TB, s, src, E, (int)n, w.hkey, w.kscr);
        tb = w.temp_bytes;
        HIP_OK(rocprim::inclusive_scan(w.temp, tb, (const int *)w.kscr, w.rb, (size_t)E, CarryKey(), s, false), "key carry scan");
        GNN_LAUNCH("pb_run_flags", pb_run_flags, gs(E + 1), TB, s, E, w.rb, w.mscan);
    }                                     // (graph-local form: pb_graph_levels wrote the run flags)
    tb = w.temp_bytes;
    HIP_OK(rocprim::exclusive_scan(w.temp, tb, (const int *)w.mscan, w.c_scan, 0, (size_t)E + 1, rocprim::plus<int>(), s, false),
           "run scan");
    GNN_LAUNCH("pb_compact", pb_compact, gs(E + 1), TB, s, E, w.mscan, w.c_scan, w.rb, w.hdr, (int)H_NRUNS);
    if (fast) {
        GNN_LAUNCH("pb_marks_fast", pb_marks_fast, 1, 1024, s, w.rb, E, (int)chunk_segments, w.marks, b.m_max, w.hdr);
    } else {
        GNN_LAUNCH("pb_mark_flags", pb_mark_flags, gs(E + 1), TB, s, w.rb, E, (int)chunk_segments, w.mscan, w.hdr);
        tb = w.temp_bytes;
        HIP_OK(rocprim::exclusive_scan(w.temp, tb, (const int *)w.mscan, w.c_scan, 0, (size_t)E + 1, rocprim::plus<int>(), s,
                                       false), "mark scan");
        GNN_LAUNCH("pb_marks", pb_marks, gs(E + 1), TB, s, w.rb, E, w.mscan, w.c_scan, w.marks, b.m_max, w.hdr);
    }
    GNN_LAUNCH("pb_chunk_bounds", pb_chunk_bounds, 1, 1024, s, w.marks, (int)chunk_segments, w.parts, w.pscan, w.cb, b.c_max,
               w.hdr);
    GNN_LAUNCH("pb_chunk_windows", pb_chunk_windows, 2048, TB, s, w.cb, src, w.src_new, w.dst_new, (int)edge_records, w.c_desc,
               w.hdr);
    GNN_LAUNCH("pb_sizes", pb_sizes, 1, 64, s, w.hdr, w.off4, sizes_out, b.np_max, b.ns_max);
    return 0;
}

int gnn_plan_build_sizes(const int32_t *src, const int32_t *dst, const int64_t *hit_ptr, int64_t n_hits,
                         int64_t n_segments, int64_t n_graphs, int32_t tile_hits, int32_t iter_records,
                         int32_t chunk_segments, int32_t edge_records, void *workspace, size_t workspace_bytes,
                         gnn_plan_sizes_t *sizes_out, void *stream)
{
    return plan_sizes_impl(src, dst, hit_ptr, nullptr, 0, 0, n_hits, n_segments, n_graphs, tile_hits, iter_records,
                           chunk_segments, edge_records, workspace, workspace_bytes, sizes_out, stream);
}

int gnn_plan_build_sizes_graphs(const int32_t *src, const int32_t *dst, const int64_t *hit_ptr, const int64_t *seg_ptr,
                                int64_t max_graph_hits, int64_t max_graph_segments, int64_t n_hits, int64_t n_segments,
                                int64_t n_graphs, int32_t tile_hits, int32_t iter_records, int32_t chunk_segments,
                                int32_t edge_records, void *workspace, size_t workspace_bytes, gnn_plan_sizes_t *sizes_out,
                                void *stream)
{
    if (!seg_ptr) return fail(GNN_ERR_BADARG, "gnn_plan_build_sizes_graphs: seg_ptr missing");
    return plan_sizes_impl(src, dst, hit_ptr, seg_ptr, max_graph_hits, max_graph_segments, n_hits, n_segments, n_graphs,
                           tile_hits, iter_records, chunk_segments, edge_records, workspace, workspace_bytes, sizes_out, stream);
}

int gnn_plan_build_fill(const float *X, int32_t F, const int32_t *src, const int32_t *dst, int64_t n_hits,
                        int64_t n_segments, int32_t chunk_segments, const gnn_plan_sizes_t *sz, void *workspace,
                        size_t workspace_bytes, const gnn_plan_out_t *out, void *stream)
{
    const int64_t n = n_hits, E = n_segments;
    if (!X || !src || !dst || !sz || !workspace || !out || n <= 0 || E <= 0 || F <= 0 || F > 16)
        return fail(GNN_ERR_BADARG, "gnn_plan_build_fill: bad argument");
    if (sz->status) return fail(GNN_ERR_UNSUPPORTED, "gnn_plan_build_fill: stage 1 reported status %lld", (long long)sz->status);
    if (workspace_bytes < gnn_plan_build_workspace_bytes(n, E, chunk_segments))
        return fail(GNN_ERR_WORKSPACE, "gnn_plan_build_fill: workspace too small");
    if (!out->X || !out->x_absmax || !out->src || !out->dst || !out->sd16 || !out->in_off || !out->in_nbr || !out->out_off ||
        !out->out_nbr || !out->in_off16 || !out->in_nbr16 || !out->out_off16 || !out->out_nbr16 || !out->tiles ||
        !out->chunks || !out->sched_a || !out->sched_b || !out->perm)
        return fail(GNN_ERR_BADARG, "gnn_plan_build_fill: output array missing");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *wb = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const Ws w = carve(wb, n, E, chunk_segments);
    const int64_t n_pad = sz->n_pad, ns = sz->n_slices;
    const int nt = (int)sz->n_tiles, nc = (int)sz->n_chunks;
    HIP_OK(hipMemsetAsync(out->X, 0, (size_t)(n_pad + 1 + 64) * F * sizeof(float), s), "memset X");
    HIP_OK(hipMemsetAsync(out->x_absmax, 0, (size_t)F * sizeof(float), s), "memset absmax");
    HIP_OK(hipMemsetAsync(out->perm, 0xFF, (size_t)n_pad * sizeof(int), s), "memset perm");
    HIP_OK(hipMemsetAsync(out->sched_a, 0xFF, (size_t)sz->n_sched * sizeof(int), s), "memset sched");
    HIP_OK(hipMemsetAsync(out->sched_b, 0xFF, (size_t)sz->n_sched * sizeof(int), s), "memset sched");
    HIP_OK(hipMemsetAsync(out->in_nbr + sz->in_total, 0, 4 * SLICE * sizeof(int), s), "memset tail");
    HIP_OK(hipMemsetAsync(out->out_nbr + sz->out_total, 0, 4 * SLICE * sizeof(int), s), "memset tail");
    HIP_OK(hipMemsetAsync(out->in_nbr16 + sz->in16_words, 0, 64 * sizeof(int), s), "memset tail");
    HIP_OK(hipMemsetAsync(out->out_nbr16 + sz->out16_words, 0, 64 * sizeof(int), s), "memset tail");
    GNN_LAUNCH("pb_fill_hits", (pb_fill_hits<16>), gs(n) < 1024u ? gs(n) : 1024u, TB, s, n, (int)F, X, w.oor, w.inv, out->X, out->perm,
               reinterpret_cast<unsigned *>(out->x_absmax));
    GNN_LAUNCH("pb_fill_offsets", pb_fill_offsets, gs(ns + 1), TB, s, ns, w.off4, out->in_off, out->out_off, out->in_off16,
               out->out_off16);
    GNN_LAUNCH("pb_fill_lists", (pb_fill_lists<true>), gs(n_pad), TB, s, n_pad, w.degn, w.ptr, w.sl4, w.off4, w.sv_in, w.t_desc,
               w.slice_tile, out->in_nbr, out->in_nbr16);
    GNN_LAUNCH("pb_fill_lists", (pb_fill_lists<false>), gs(n_pad), TB, s, n_pad, w.degn, w.ptr, w.sl4, w.off4, w.sv_out,
               w.t_desc, w.slice_tile, out->out_nbr, out->out_nbr16);
    if (nt > 0) {
        GNN_LAUNCH("pb_copy_i32", pb_copy_i32, gs((int64_t)nt * 8), TB, s, w.t_desc, out->tiles, (int64_t)nt * 8);
        GNN_LAUNCH("pb_schedule", pb_schedule, (unsigned)(nt < 4096 ? nt : 4096), 64, s, nt, w.t_desc, w.sl4, out->sched_a,
                   out->sched_b);
    }
    if (nc > 0)
        GNN_LAUNCH("pb_fill_chunks", pb_fill_chunks, (unsigned)(nc < 4096 ? nc : 4096), TB, s, nc, w.c_desc, src, w.src_new,
                   w.dst_new, out->chunks, out->src, out->dst, out->sd16);
    if (out->src_abs) GNN_LAUNCH("pb_copy_i32", pb_copy_i32, gs(E), TB, s, w.src_new, out->src_abs, E);
    if (out->dst_abs) GNN_LAUNCH("pb_copy_i32", pb_copy_i32, gs(E), TB, s, w.dst_new, out->dst_abs, E);
    if (out->level) GNN_LAUNCH("pb_copy_i32", pb_copy_i32, gs(n), TB, s, w.lvA, out->level, n);
    return 0;
}

}  // extern "C"
