// Track ids and the track-point table of a whole sequence on the device (gfx950).
//
// The reference's track store hands out track ids row by row (models/TrackDatabase.py:34-76,
// 166-184): an inlier row (prev_left, prev_right, cur_left, cur_right) of pair p continues the track
// that the LAST row of pair p-1 with cur_left == prev_left belongs to (the parity-slot dict keeps the
// last writer), otherwise it opens a new track whose id is the number of tracks opened before it, in
// (pair, row) order.  That is a forest over the rows: parent(g) = the row it continues.  Here
//
//   link      last row per (pair, cur_left keypoint)                       atomicMax scatter
//   parent    parent(g) from the previous pair's links; roots = new tracks
//   jump      pointer jumping with distances: every row learns its root and its depth
//   scan      exclusive scan of the root flags = track id of every root (ids in arrival order)
//   ids       track id of a row = id of its root; points per track = deepest row + 2
//   scan      exclusive scan of the points per track = first point slot of every track
//   points    row g's later-frame point goes to slot start[track] + 1 + depth(g), a root's
//             earlier-frame point to start[track]
//
// so the host receives the track store as flat arrays (CSR by track) and never replays rows.
// A row has at most one child whenever prev_left is unique within a pair -- true for everything the
// front-end emits (the temporal matcher keeps one match per query keypoint).  Tables where two rows
// of a pair share prev_left make a tree instead of chains (the reference then appends both rows to
// one track); that is detected and reported in the summary, and the caller replays on the host.
#include "common.h"

namespace slam {
namespace {

constexpr int SCAN_ITEMS = 8;                   // per thread
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;     // per block

// summary slots
enum { S_ROWS = 0, S_TRACKS = 1, S_POINTS = 2, S_MAXLEN = 3, S_FLAGS = 4 };
enum { FLAG_BRANCH = 1, FLAG_KP_RANGE = 2, FLAG_ROW_OVERFLOW = 4 };

__device__ __forceinline__ unsigned long long pack_node(int parent, int depth)
{
    return ((unsigned long long)(unsigned)parent << 32) | (unsigned long long)(unsigned)depth;
}

// Exclusive scan of the rows per pair (one block; n_pairs is a few thousand) and the totals.
__global__ void __launch_bounds__(256) track_offsets_kernel(const int32_t *__restrict__ n_inl, int n_pairs, int cap_rows,
                                                            int32_t *__restrict__ off, int32_t *__restrict__ summary)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_pairs; base += 256) {
        const int p = base + threadIdx.x;
        const int v = p < n_pairs ? max(n_inl[p], 0) : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d)
                incl += o;
        }
        if (lane == 63)
            wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w)
            before += wsum[w];
        if (p < n_pairs)
            off[p] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0)
            carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int total = carry, flags = 0;
        if (total > cap_rows) {             // the tables cannot hold what the counts say: report, work on nothing
            flags |= FLAG_ROW_OVERFLOW;
            total = 0;
        }
        off[n_pairs] = total;
        summary[S_ROWS] = total;
        summary[S_TRACKS] = 0;
        summary[S_POINTS] = 0;
        summary[S_MAXLEN] = 0;
        summary[S_FLAGS] = flags;
        summary[5] = summary[6] = summary[7] = 0;
    }
}

// link[p * kp_cap + cur_left] = last row of pair p that ends on that keypoint
__global__ void __launch_bounds__(256) track_link_kernel(const int4 *__restrict__ rows, const int32_t *__restrict__ off,
                                                         int kp_cap, int32_t *__restrict__ link,
                                                         int32_t *__restrict__ summary)
{
    const int p = blockIdx.x;
    if (summary[S_ROWS] == 0)
        return;
    const int g0 = off[p], g1 = off[p + 1];
    for (int g = g0 + threadIdx.x; g < g1; g += 256) {
        const int c = rows[g].z;
        if ((unsigned)c < (unsigned)kp_cap)
            atomicMax(&link[(size_t)p * kp_cap + c], g);
        else
            atomicOr(&summary[S_FLAGS], FLAG_KP_RANGE);
    }
}

// parent of every row; children per row are counted in `child` (zeroed) to detect trees
__global__ void __launch_bounds__(256) track_parent_kernel(const int4 *__restrict__ rows, const int32_t *__restrict__ off,
                                                           int kp_cap, const int32_t *__restrict__ link,
                                                           unsigned long long *__restrict__ node,
                                                           int32_t *__restrict__ child, int32_t *__restrict__ summary)
{
    const int p = blockIdx.x;
    if (summary[S_ROWS] == 0)
        return;
    const int g0 = off[p], g1 = off[p + 1];
    for (int g = g0 + threadIdx.x; g < g1; g += 256) {
        const int a = rows[g].x;
        int c = -1;
        if ((unsigned)a >= (unsigned)kp_cap)
            atomicOr(&summary[S_FLAGS], FLAG_KP_RANGE);
        else if (p > 0)
            c = link[(size_t)(p - 1) * kp_cap + a];
        if (c >= 0) {
            node[g] = pack_node(c, 1);
            if (atomicAdd(&child[c], 1) > 0)
                atomicOr(&summary[S_FLAGS], FLAG_BRANCH);
        } else {
            node[g] = pack_node(g, 0);
        }
    }
}

// One round of pointer jumping, in place.  node[x] = (a, d) always means "a is an ancestor of x, d
// links up"; a row's entry is written only by its own thread and its parent's entry is read with one
// 64-bit load, so any interleaving of rounds keeps that invariant.
__global__ void __launch_bounds__(256) track_jump_kernel(unsigned long long *node,
                                                         const int32_t *__restrict__ summary)
{
    const int n = summary[S_ROWS];
    for (int g = blockIdx.x * 256 + threadIdx.x; g < n; g += gridDim.x * 256) {
        const unsigned long long me = node[g];
        const int par = (int)(me >> 32);
        if (par == g)
            continue;
        const unsigned long long up = __hip_atomic_load(&node[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pp = (int)(up >> 32);
        if (pp != par)
            node[g] = pack_node(pp, (int)(unsigned)me + (int)(unsigned)up);
    }
}

// ---- exclusive scan of a device-side sequence f(0..n) ------------------------------------------
struct RootFlag {          // 1 where row i opens a track
    const unsigned long long *node;
    const int32_t *summary;
    __device__ int count() const { return summary[S_ROWS]; }
    __device__ int operator()(int i) const { return (int)(node[i] >> 32) == i ? 1 : 0; }
};
struct TrackPoints {       // points of track i
    const int32_t *len;
    const int32_t *summary;
    __device__ int count() const { return summary[S_TRACKS]; }
    __device__ int operator()(int i) const { return len[i]; }
};

template <typename F>
__global__ void __launch_bounds__(256) scan_partial_kernel(F f, int32_t *__restrict__ partials)
{
    __shared__ int red[4];
    const int n = f.count();
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; ++e)
        s += (base + e < n) ? f(base + e) : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the block totals in place (one block), grand total to *total
__global__ void __launch_bounds__(256) scan_partials_kernel(int32_t *__restrict__ partials, int n_blocks,
                                                            int32_t *__restrict__ total)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_blocks; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < n_blocks ? partials[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d)
                incl += o;
        }
        if (lane == 63)
            wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w)
            before += wsum[w];
        if (i < n_blocks)
            partials[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0)
            carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *total = carry;
}

// out[i] = sum of f(j), j < i, for i < n; out[n] = the total when `closed`
template <typename F>
__global__ void __launch_bounds__(256) scan_final_kernel(F f, const int32_t *__restrict__ partials,
                                                         int32_t *__restrict__ out, bool closed)
{
    __shared__ int wsum[4];
    const int n = f.count();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    if ((int)blockIdx.x * SCAN_TILE > n)
        return;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; ++e) {
        v[e] = (base + e < n) ? f(base + e) : 0;
        s += v[e];
    }
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d)
            incl += o;
    }
    if (lane == 63)
        wsum[wave] = incl;
    __syncthreads();
    int run = partials[blockIdx.x] + incl - s;
    for (int w = 0; w < wave; ++w)
        run += wsum[w];
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; ++e) {
        if (base + e < n || (closed && base + e == n))
            out[base + e] = run;
        run += v[e];
    }
}

// track id and depth of every row; points per track = deepest row + 2
__global__ void __launch_bounds__(256) track_id_kernel(const unsigned long long *__restrict__ node,
                                                       const int32_t *__restrict__ root_rank,
                                                       int32_t *__restrict__ row_track, int32_t *__restrict__ row_depth,
                                                       int32_t *__restrict__ track_len, int32_t *__restrict__ summary)
{
    const int n = summary[S_ROWS];
    int longest = 0;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < n; g += gridDim.x * 256) {
        const unsigned long long me = node[g];
        const int root = (int)(me >> 32), depth = (int)(unsigned)me;
        const int t = root_rank[root];
        row_track[g] = t;
        row_depth[g] = depth;
        atomicMax(&track_len[t], depth + 2);
        longest = max(longest, depth + 2);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        longest = max(longest, __shfl_xor(longest, d));
    if ((threadIdx.x & 63) == 0 && longest > 0)
        atomicMax(&summary[S_MAXLEN], longest);
}

// the track-point table: (left keypoint, frame, (xl, xr, yl)) per point, tracks back to back
__global__ void __launch_bounds__(256) track_points_kernel(const int4 *__restrict__ rows, const float *__restrict__ loc,
                                                           const int32_t *__restrict__ off, int first_frame,
                                                           const int32_t *__restrict__ row_track,
                                                           const int32_t *__restrict__ row_depth,
                                                           const int32_t *__restrict__ track_start,
                                                           int32_t *__restrict__ pt_kp, int32_t *__restrict__ pt_frame,
                                                           float *__restrict__ pt_loc, const int32_t *__restrict__ summary)
{
    const int p = blockIdx.x;
    if (summary[S_ROWS] == 0)
        return;
    const int g0 = off[p], g1 = off[p + 1];
    for (int g = g0 + threadIdx.x; g < g1; g += 256) {
        const int4 r = rows[g];
        const float *l = loc + 6 * (size_t)g;
        const int depth = row_depth[g];
        const int s0 = track_start[row_track[g]];
        const int s = s0 + 1 + depth;
        pt_kp[s] = r.z;
        pt_frame[s] = first_frame + p + 1;
        pt_loc[3 * (size_t)s + 0] = l[3];
        pt_loc[3 * (size_t)s + 1] = l[4];
        pt_loc[3 * (size_t)s + 2] = l[5];
        if (depth == 0) {
            pt_kp[s0] = r.x;
            pt_frame[s0] = first_frame + p;
            pt_loc[3 * (size_t)s0 + 0] = l[0];
            pt_loc[3 * (size_t)s0 + 1] = l[1];
            pt_loc[3 * (size_t)s0 + 2] = l[2];
        }
    }
}

struct WorkLayout {
    size_t off, link, node, rank, len, partials, total;
    int n_blocks;
};

WorkLayout work_layout(int n_pairs, int cap_rows, int kp_cap)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WorkLayout w;
    size_t o = 0;
    w.off = o;
    o = up(o + sizeof(int32_t) * ((size_t)n_pairs + 1));
    w.link = o;
    o = up(o + sizeof(int32_t) * (size_t)n_pairs * (size_t)kp_cap);
    w.node = o;
    o = up(o + sizeof(unsigned long long) * (size_t)cap_rows);
    w.rank = o;
    o = up(o + sizeof(int32_t) * ((size_t)cap_rows + 1));
    w.len = o;
    o = up(o + sizeof(int32_t) * ((size_t)cap_rows + 1));
    w.n_blocks = (cap_rows + 1 + SCAN_TILE - 1) / SCAN_TILE;
    w.partials = o;
    o = up(o + sizeof(int32_t) * (size_t)w.n_blocks);
    w.total = o;
    return w;
}

}  // namespace
}  // namespace slam

using namespace slam;

extern "C" size_t slam_track_assign_work_bytes(int n_pairs, int cap_rows, int kp_cap)
{
    if (n_pairs < 0 || cap_rows < 0 || kp_cap < 0)
        return 0;
    return work_layout(n_pairs, cap_rows, kp_cap).total;
}

extern "C" int slam_track_assign(const int32_t *inl_rows, const float *inl_loc, const int32_t *n_inl, int n_pairs,
                                 int cap_rows, int kp_cap, int first_frame, int32_t *row_track, int32_t *row_depth,
                                 int32_t *track_start, int32_t *pt_kp, int32_t *pt_frame, float *pt_loc,
                                 int32_t *summary, void *work, size_t work_bytes, void *stream)
{
    if (n_pairs < 0 || cap_rows < 0 || kp_cap < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_track_assign: negative size");
    if (!summary)
        return fail(SLAM_ERR_BAD_ARG, "slam_track_assign: null summary");
    hipStream_t s = (hipStream_t)stream;
    if (n_pairs == 0 || cap_rows == 0) {
        hipError_t e = hipMemsetAsync(summary, 0, 8 * sizeof(int32_t), s);
        if (e != hipSuccess)
            return fail(SLAM_ERR_LAUNCH, "slam_track_assign: %s", hipGetErrorString(e));
        return SLAM_OK;
    }
    if (!inl_rows || !inl_loc || !n_inl || !row_track || !row_depth || !track_start || !pt_kp || !pt_frame || !pt_loc ||
        !work)
        return fail(SLAM_ERR_BAD_ARG, "slam_track_assign: null buffer");
    if ((reinterpret_cast<uintptr_t>(inl_rows) & 15) || (reinterpret_cast<uintptr_t>(work) & 255))
        return fail(SLAM_ERR_BAD_ARG, "slam_track_assign: rows must be 16-byte and work 256-byte aligned");
    if ((long long)n_pairs * kp_cap > 0x7FFFFFFFLL || cap_rows > 0x3FFFFFFF)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_track_assign: tables too large for 32-bit slots");
    const WorkLayout w = work_layout(n_pairs, cap_rows, kp_cap);
    if (work_bytes < w.total)
        return fail(SLAM_ERR_BAD_ARG, "slam_track_assign: work_bytes %zu < %zu (slam_track_assign_work_bytes)", work_bytes,
                    w.total);
    char *wb = static_cast<char *>(work);
    int32_t *off = reinterpret_cast<int32_t *>(wb + w.off);
    int32_t *link = reinterpret_cast<int32_t *>(wb + w.link);
    unsigned long long *node = reinterpret_cast<unsigned long long *>(wb + w.node);
    int32_t *rank = reinterpret_cast<int32_t *>(wb + w.rank);
    int32_t *len = reinterpret_cast<int32_t *>(wb + w.len);
    int32_t *partials = reinterpret_cast<int32_t *>(wb + w.partials);
    const int4 *rows4 = reinterpret_cast<const int4 *>(inl_rows);

    hipError_t e = hipMemsetAsync(link, 0xFF, sizeof(int32_t) * (size_t)n_pairs * kp_cap, s);
    if (e == hipSuccess)
        e = hipMemsetAsync(row_track, 0, sizeof(int32_t) * (size_t)cap_rows, s);      // child counters first
    if (e == hipSuccess)
        e = hipMemsetAsync(len, 0, sizeof(int32_t) * ((size_t)cap_rows + 1), s);
    if (e != hipSuccess)
        return fail(SLAM_ERR_LAUNCH, "slam_track_assign: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(track_offsets_kernel, dim3(1), dim3(256), 0, s, n_inl, n_pairs, cap_rows, off, summary);
    hipLaunchKernelGGL(track_link_kernel, dim3(n_pairs), dim3(256), 0, s, rows4, off, kp_cap, link, summary);
    hipLaunchKernelGGL(track_parent_kernel, dim3(n_pairs), dim3(256), 0, s, rows4, off, kp_cap, link, node, row_track,
                       summary);
    const int row_blocks = min((cap_rows + 255) / 256, 8192);
    int rounds = 1;
    while ((1 << rounds) < n_pairs)
        ++rounds;
    for (int r = 0; r < rounds; ++r)
        hipLaunchKernelGGL(track_jump_kernel, dim3(row_blocks), dim3(256), 0, s, node, summary);
    const RootFlag roots{node, summary};
    hipLaunchKernelGGL(scan_partial_kernel<RootFlag>, dim3(w.n_blocks), dim3(256), 0, s, roots, partials);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(256), 0, s, partials, w.n_blocks, summary + S_TRACKS);
    hipLaunchKernelGGL(scan_final_kernel<RootFlag>, dim3(w.n_blocks), dim3(256), 0, s, roots, partials, rank, false);
    hipLaunchKernelGGL(track_id_kernel, dim3(row_blocks), dim3(256), 0, s, node, rank, row_track, row_depth, len, summary);
    const TrackPoints points{len, summary};
    hipLaunchKernelGGL(scan_partial_kernel<TrackPoints>, dim3(w.n_blocks), dim3(256), 0, s, points, partials);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(256), 0, s, partials, w.n_blocks, summary + S_POINTS);
    hipLaunchKernelGGL(scan_final_kernel<TrackPoints>, dim3(w.n_blocks), dim3(256), 0, s, points, partials, track_start,
                       true);
    hipLaunchKernelGGL(track_points_kernel, dim3(n_pairs), dim3(256), 0, s, rows4, inl_loc, off, first_frame, row_track,
                       row_depth, track_start, pt_kp, pt_frame, pt_loc, summary);
    return check_launch("slam_track_assign");
}
