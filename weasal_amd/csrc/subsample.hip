// weasal_amd/csrc/subsample.hip -- voxel-grid barycentre subsampling on gfx950.
// (compiled with -ffp-contract=off; every f32 operation rounds exactly like the reference's)
//
// Replaces grid_subsampling / batch_grid_subsampling
// (cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106, :109-211).
// The reference walks the points once, keeping an std::unordered_map<size_t, SampledData>.
// Here, per batch element:
//   1. min/max reduction -> origin = floor(min * (1/dl)) * dl, nX, nY            (:25-31)
//   2. cell key per point, IEEE f32 divide + floor                               (:53-56)
//   3. keys are inserted into an open-addressing hash table in HBM (64-bit CAS); per slot the
//      point count and the smallest point index (= first occurrence) are kept with atomics
//   4. cells are ranked by first occurrence (mark + prefix sum) and the point lists of the
//      cells are built by a counting sort and sorted by point index, so that
//   5. the per-cell sums run SEQUENTIALLY in input order -- bit-identical to the reference's
//      running f32 sums (:61-68) -- and the barycentre is sum * (float)(1.0 / count)   (:87)
//   6. WS_ORDER_REFERENCE: rows are emitted in the iteration order the reference's
//      std::unordered_map would have (:85).  libstdc++'s _Hashtable keeps one forward list; a
//      node whose bucket is empty is linked at the list head, otherwise right behind its
//      bucket's "before" node.  Filling an empty table of B buckets with a sequence S therefore
//      yields the order "buckets by first use, latest first; inside a bucket latest first", and a
//      rehash (bucket counts 13, 29, 59, ... from std::__detail::_Prime_rehash_policy, growth when
//      size exceeds the bucket count) replays the current list into the larger table.  Each phase is
//      a parallel counting sort; one 1024-thread workgroup per batch element runs all phases.
#include "ws_scan.h"
#include <unordered_map>
#include <vector>

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;

struct SubCloud {
    int base, len;                 // points of this element
    int tab_base, tab_mask;        // hash table region (power of two)
    float org[3];
    unsigned long long nX, nY;
    int rank_base, m;              // cells of this element: ranks [rank_base, rank_base + m)
    int out_len, out_base;         // rows kept after max_p, first output row
    int ord_base, ord_bcap;        // scratch region of the order kernel
    float mn[3], mx[3];
};

__global__ __launch_bounds__(1024) void sub_bbox_kernel(const float* __restrict__ pts, SubCloud* __restrict__ clouds)
{
    __shared__ float red[6][16];
    SubCloud& g = clouds[blockIdx.x];
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = threadIdx.x; i < g.len; i += blockDim.x) {
        const float* p = pts + 3 * (int64_t)(g.base + i);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], p[d]); mx[d] = fmaxf(mx[d], p[d]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64));
        }
        if (lane == 0) { red[d][wave] = mn[d]; red[3 + d][wave] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && g.len > 0) {
        const float dl = g.org[0];   // dl parked here by the host
        for (int d = 0; d < 3; ++d) {
            float a = red[d][0], b = red[3 + d][0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { a = fminf(a, red[d][w]); b = fmaxf(b, red[3 + d][w]); }
            g.mn[d] = a; g.mx[d] = b;
        }
        const float inv = 1.0f / dl;                                        // (1/sampleDl), :27
        float org[3];
        for (int d = 0; d < 3; ++d) org[d] = floorf(g.mn[d] * inv) * dl;     // :27
        g.nX = (unsigned long long)(long long)floorf(__fdiv_rn(g.mx[0] - org[0], dl)) + 1ull;   // :30
        g.nY = (unsigned long long)(long long)floorf(__fdiv_rn(g.mx[1] - org[1], dl)) + 1ull;   // :31
        g.org[0] = org[0]; g.org[1] = org[1]; g.org[2] = org[2];
    }
}

__device__ __forceinline__ int cloud_of_point(const SubCloud* __restrict__ clouds, int nb, int64_t i)
{
    int b = 0;
    while (b + 1 < nb && i >= clouds[b].base + clouds[b].len) ++b;
    return b;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

__global__ __launch_bounds__(256) void sub_insert_kernel(const float* __restrict__ pts, int64_t n,
                                                          const SubCloud* __restrict__ clouds, int nb, float dl,
                                                          unsigned long long* __restrict__ tab_key,
                                                          int32_t* __restrict__ tab_cnt, int32_t* __restrict__ tab_first,
                                                          int32_t* __restrict__ slot_of)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = cloud_of_point(clouds, nb, i);
        const SubCloud& g = clouds[b];
        const float* p = pts + 3 * i;
        // (size_t)floor(...) of the reference (x86-64: cvttss2si, negative values wrap)
        const unsigned long long iX = (unsigned long long)(long long)floorf(__fdiv_rn(p[0] - g.org[0], dl));
        const unsigned long long iY = (unsigned long long)(long long)floorf(__fdiv_rn(p[1] - g.org[1], dl));
        const unsigned long long iZ = (unsigned long long)(long long)floorf(__fdiv_rn(p[2] - g.org[2], dl));
        unsigned long long key = iX + g.nX * iY + g.nX * g.nY * iZ;           // :56
        if (key == EMPTY_KEY) key = EMPTY_KEY - 1;                            // reserved value
        unsigned h = (unsigned)mix64(key) & (unsigned)g.tab_mask;
        for (;;) {
            const int slot = g.tab_base + (int)h;
            const unsigned long long old = atomicCAS(&tab_key[slot], EMPTY_KEY, key);
            if (old == EMPTY_KEY || old == key) {
                atomicAdd(&tab_cnt[slot], 1);
                atomicMin(&tab_first[slot], (int)i);
                slot_of[i] = slot;
                break;
            }
            h = (h + 1) & (unsigned)g.tab_mask;
        }
    }
}

// mark[first occurrence point] = 1 ; first_slot[point] = slot
__global__ __launch_bounds__(256) void sub_mark_kernel(int64_t tab_total, const int32_t* __restrict__ tab_cnt,
                                                        const int32_t* __restrict__ tab_first,
                                                        int32_t* __restrict__ flag, int32_t* __restrict__ first_slot)
{
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < tab_total; s += (int64_t)gridDim.x * 256) {
        if (tab_cnt[s] > 0) {
            const int f = tab_first[s];
            flag[f] = 1;
            first_slot[f] = (int)s;
        }
    }
}

// after the scan of flag -> rank[]: per first-occurrence point write the cell tables (by rank)
__global__ __launch_bounds__(256) void sub_cells_kernel(int64_t n, const int32_t* __restrict__ rank /*[n+1]*/,
                                                         const int32_t* __restrict__ first_slot,
                                                         const unsigned long long* __restrict__ tab_key,
                                                         const int32_t* __restrict__ tab_cnt,
                                                         int32_t* __restrict__ slot_rank, int32_t* __restrict__ cell_cnt,
                                                         unsigned long long* __restrict__ cell_key)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (rank[i + 1] != rank[i]) {
            const int r = rank[i], s = first_slot[i];
            slot_rank[s] = r;
            cell_cnt[r] = tab_cnt[s];
            cell_key[r] = tab_key[s];
        }
    }
}

__global__ void sub_cloud_ranks_kernel(SubCloud* __restrict__ clouds, int nb, const int32_t* __restrict__ rank, int max_p)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    SubCloud& g = clouds[b];
    g.rank_base = rank[g.base];
    g.m = rank[g.base + g.len] - g.rank_base;
    g.out_len = (max_p > 0 && g.m > max_p) ? max_p : g.m;    // grid_subsampling.cpp:133-134,181-204
}

__global__ __launch_bounds__(256) void sub_list_fill_kernel(int64_t n, const int32_t* __restrict__ slot_of,
                                                             const int32_t* __restrict__ slot_rank,
                                                             int32_t* __restrict__ cursor, int32_t* __restrict__ list)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = slot_rank[slot_of[i]];
        list[atomicAdd(&cursor[r], 1)] = (int)i;
    }
}

// one thread per cell: insertion sort of its (short) point list -> input order
__global__ __launch_bounds__(256) void sub_list_sort_kernel(const int32_t* __restrict__ total_cells,
                                                             const int32_t* __restrict__ cell_off, int32_t* __restrict__ list)
{
    const int M = *total_cells;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < M; r += gridDim.x * 256) {
        const int beg = cell_off[r], end = cell_off[r + 1];
        for (int a = beg + 1; a < end; ++a) {
            const int v = list[a];
            int b = a - 1;
            while (b >= beg && list[b] > v) { list[b + 1] = list[b]; --b; }
            list[b + 1] = v;
        }
    }
}

// ---- reference row order (see the file header) -------------------------------------------------
// block-wide inclusive scan of arr[0..n) in place (global memory), 1024 threads
__device__ void block_inclusive_scan_inplace(int* arr, int n, int* lds /*[17]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        int v = i < n ? arr[i] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v += t;
        }
        if (lane == 63) lds[wave] = v;
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int s = lds[w]; if (w < wave) off += s; tot += s; }
        if (i < n) arr[i] = v + off + carry;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void sub_order_kernel(const SubCloud* __restrict__ clouds,
                                                          const unsigned long long* __restrict__ cell_key,
                                                          int32_t* __restrict__ out_pos, int32_t* __restrict__ scratch,
                                                          const unsigned long long* __restrict__ primes, int nprimes,
                                                          int order_mode)
{
    __shared__ int lds[17];
    const SubCloud g = clouds[blockIdx.x];
    const int m = g.m, rb = g.rank_base;
    if (m <= 0) return;
    if (order_mode == WS_ORDER_FIRST_SEEN) {
        for (int i = threadIdx.x; i < m; i += 1024) out_pos[rb + i] = i;
        return;
    }
    // scratch layout for this element (ints): seqA[len] seqB[len] bkt[len] memb[len] pre[len] act[bcap] cnt[bcap+1] fil[bcap]
    int* seqA = scratch + g.ord_base;
    int* seqB = seqA + g.len;
    int* bkt = seqB + g.len;
    int* memb = bkt + g.len;
    int* pre = memb + g.len;
    int* act = pre + g.len;
    int* cnt = act + g.ord_bcap;
    int* fil = cnt + g.ord_bcap + 1;
    int* cur = seqA;
    int* nxt = seqB;
    int nprev = 0;
    for (int j = 0; j < nprimes && nprev < m; ++j) {
        const unsigned long long B = primes[j];
        const int n = (unsigned long long)m < B ? m : (int)B;   // elements present when this phase ends
        const int Bi = (int)B;
        for (int b = threadIdx.x; b < Bi; b += 1024) { act[b] = 0x7fffffff; cnt[b] = 0; fil[b] = 0; }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int id = i < nprev ? cur[i] : i;
            const int bk = (int)(cell_key[rb + id] % B);
            bkt[i] = bk;
            atomicMin(&act[bk], i);
            atomicAdd(&cnt[bk], 1);
        }
        __syncthreads();
        // act/cnt were produced by L2 atomics: re-read them past the L1 (agent-scope loads) and
        // store them back so that the plain loads below cannot hit a stale L1 line
        for (int b = threadIdx.x; b < Bi; b += 1024) {
            const int a = __hip_atomic_load(&act[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int c = __hip_atomic_load(&cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            act[b] = a; cnt[b] = c;
        }
        __syncthreads();
        // pre[i] = inclusive prefix of A[i] = (act[bkt[i]] == i) ? cnt[bkt[i]] : 0
        for (int i = threadIdx.x; i < n; i += 1024) pre[i] = (act[bkt[i]] == i) ? cnt[bkt[i]] : 0;
        __syncthreads();
        block_inclusive_scan_inplace(pre, n, lds);
        // cnt -> inclusive bucket offsets (members of bucket b live in [cnt[b]-size, cnt[b]))
        block_inclusive_scan_inplace(cnt, Bi, lds);
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int bk = bkt[i];
            const int size = cnt[bk] - (bk ? cnt[bk - 1] : 0);
            const int beg = cnt[bk] - size;
            memb[beg + atomicAdd(&fil[bk], 1)] = i;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int bk = bkt[i];
            const int size = cnt[bk] - (bk ? cnt[bk - 1] : 0);
            const int beg = cnt[bk] - size;
            int later = 0;
            for (int p = beg; p < beg + size; ++p) later += memb[p] > i;
            const int start = n - pre[act[bk]];      // elements of buckets first used after this one
            const int id = i < nprev ? cur[i] : i;
            nxt[start + later] = id;
        }
        __syncthreads();
        int* t = cur; cur = nxt; nxt = t;
        nprev = n;
    }
    for (int i = threadIdx.x; i < m; i += 1024) out_pos[rb + cur[i]] = i;
}

// ---- emit ----------------------------------------------------------------------------------------
__device__ __forceinline__ int cloud_of_rank(const SubCloud* __restrict__ clouds, int nb, int r)
{
    int b = 0;
    while (b + 1 < nb && r >= clouds[b].rank_base + clouds[b].m) ++b;
    return b;
}

__global__ __launch_bounds__(256) void sub_emit_points_kernel(const float* __restrict__ pts, const SubCloud* __restrict__ clouds,
                                                               int nb, int M, const int32_t* __restrict__ cell_off,
                                                               const int32_t* __restrict__ list, const int32_t* __restrict__ out_pos,
                                                               const unsigned long long* __restrict__ cell_key,
                                                               float* __restrict__ out_points, unsigned long long* __restrict__ out_keys,
                                                               int32_t* __restrict__ out_counts)
{
    for (int r = blockIdx.x * 256 + threadIdx.x; r < M; r += gridDim.x * 256) {
        const int b = cloud_of_rank(clouds, nb, r);
        const SubCloud& g = clouds[b];
        const int pos = out_pos[r];
        if (pos >= g.out_len) continue;
        const int row = g.out_base + pos;
        const int beg = cell_off[r], end = cell_off[r + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int p = beg; p < end; ++p) {                   // input order: sequential f32 sums
            const float* q = pts + 3 * (int64_t)list[p];
            sx += q[0]; sy += q[1]; sz += q[2];
        }
        const float a = (float)(1.0 / (double)(end - beg));  // grid_subsampling.cpp:87
        out_points[3 * (int64_t)row + 0] = sx * a;
        out_points[3 * (int64_t)row + 1] = sy * a;
        out_points[3 * (int64_t)row + 2] = sz * a;
        if (out_keys) out_keys[row] = cell_key[r];
        if (out_counts) out_counts[row] = end - beg;
    }
}

__global__ __launch_bounds__(256) void sub_emit_features_kernel(const float* __restrict__ feat, int fd,
                                                                 const SubCloud* __restrict__ clouds, int nb, int M,
                                                                 const int32_t* __restrict__ cell_off, const int32_t* __restrict__ list,
                                                                 const int32_t* __restrict__ out_pos, float* __restrict__ out_f)
{
    const int64_t total = (int64_t)M * fd;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int r = (int)(t / fd), k = (int)(t - (int64_t)r * fd);
        const int b = cloud_of_rank(clouds, nb, r);
        const SubCloud& g = clouds[b];
        const int pos = out_pos[r];
        if (pos >= g.out_len) continue;
        const int beg = cell_off[r], end = cell_off[r + 1];
        float s = 0.f;
        for (int p = beg; p < end; ++p) s += feat[(int64_t)list[p] * fd + k];
        out_f[(int64_t)(g.out_base + pos) * fd + k] = __fdiv_rn(s, (float)(end - beg));   // :90-94
    }
}

// Label of a cell = first maximum of an unordered_map<int,int> histogram in iteration order
// (grid_subsampling.cpp:99-101).  The histogram's iteration order follows the same libstdc++
// insertion rule as the cell map; it is replayed here for up to MAXL distinct labels per cell
// (bucket count 13, then 29, 59, ... on growth); beyond that the first-seen label among the
// maxima is used.
constexpr int MAXL = 48;
__device__ int label_argmax(const int* labs, const int* cnts, int nl, const unsigned long long* primes)
{
    // order[] = current list order (indices into labs)
    int order[MAXL], tmp[MAXL], actv[MAXL];
    int nprev = 0;
    for (int j = 0; nprev < nl; ++j) {
        const unsigned long long B = primes[j];
        const int n = (unsigned long long)nl < B ? nl : (int)B;
        // sequence S = order[0..nprev) ++ nprev..n-1 ; bucket of element
        int seq[MAXL];
        for (int i = 0; i < n; ++i) seq[i] = i < nprev ? order[i] : i;
        // activation time of each element's bucket
        for (int i = 0; i < n; ++i) {
            const unsigned long long bi = (unsigned long long)(long long)labs[seq[i]] % B;
            int a = i;
            for (int i2 = 0; i2 < i; ++i2)
                if ((unsigned long long)(long long)labs[seq[i2]] % B == bi) { a = i2; break; }
            actv[i] = a;
        }
        // final position: sort by (activation desc, i desc)
        for (int i = 0; i < n; ++i) {
            int pos = 0;
            for (int i2 = 0; i2 < n; ++i2)
                if (actv[i2] > actv[i] || (actv[i2] == actv[i] && i2 > i)) ++pos;
            tmp[pos] = seq[i];
        }
        for (int i = 0; i < n; ++i) order[i] = tmp[i];
        nprev = n;
    }
    int best = order[0];
    for (int i = 1; i < nl; ++i)
        if (cnts[order[i]] > cnts[best]) best = order[i];
    return labs[best];
}

__global__ __launch_bounds__(64) void sub_emit_labels_kernel(const int32_t* __restrict__ labels, int ld,
                                                              const SubCloud* __restrict__ clouds, int nb, int M,
                                                              const int32_t* __restrict__ cell_off, const int32_t* __restrict__ list,
                                                              const int32_t* __restrict__ out_pos,
                                                              const unsigned long long* __restrict__ primes,
                                                              int32_t* __restrict__ out_l)
{
    const int64_t total = (int64_t)M * ld;
    for (int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 64) {
        const int r = (int)(t / ld), k = (int)(t - (int64_t)r * ld);
        const int b = cloud_of_rank(clouds, nb, r);
        const SubCloud& g = clouds[b];
        const int pos = out_pos[r];
        if (pos >= g.out_len) continue;
        const int beg = cell_off[r], end = cell_off[r + 1];
        int labs[MAXL], cnts[MAXL];
        int nl = 0;
        bool overflow = false;
        for (int p = beg; p < end; ++p) {
            const int v = labels[(int64_t)list[p] * ld + k];
            int f = -1;
            for (int a = 0; a < nl; ++a) if (labs[a] == v) { f = a; break; }
            if (f >= 0) cnts[f]++;
            else if (nl < MAXL) { labs[nl] = v; cnts[nl] = 1; ++nl; }
            else overflow = true;
        }
        int res;
        if (overflow) {
            int best = 0;
            for (int a = 1; a < nl; ++a) if (cnts[a] > cnts[best]) best = a;
            res = labs[best];
        } else {
            res = label_argmax(labs, cnts, nl, primes);
        }
        out_l[(int64_t)(g.out_base + pos) * ld + k] = res;
    }
}

__global__ __launch_bounds__(256) void rotate_kernel(const float* __restrict__ pts, int64_t n, const int32_t* __restrict__ lens,
                                                      int nb, const float* __restrict__ rot, int transpose, float* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int b = 0;
        int64_t acc = lens[0];
        while (b + 1 < nb && i >= acc) { ++b; acc += lens[b]; }
        const float* R = rot + 9 * b;
        const float p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float r0 = transpose ? R[3 * j + 0] : R[0 + j];
            const float r1 = transpose ? R[3 * j + 1] : R[3 + j];
            const float r2 = transpose ? R[3 * j + 2] : R[6 + j];
            float v = p0 * r0;
            v = v + p1 * r1;
            v = v + p2 * r2;
            out[3 * i + j] = v;
        }
    }
}

// host-side lengths and matrices travel as a kernel argument (copied at launch, no H2D staging copy
// and no synchronisation); up to ROT_MAX batch elements
constexpr int ROT_MAX = 64;
struct RotTable { int32_t end[ROT_MAX]; float r[ROT_MAX * 9]; };
__global__ __launch_bounds__(256) void rotate_table_kernel(const float* __restrict__ pts, int64_t n, RotTable t, int nb,
                                                            int transpose, float* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int b = 0;
        while (b + 1 < nb && i >= t.end[b]) ++b;
        const float* R = t.r + 9 * b;
        const float p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float r0 = transpose ? R[3 * j + 0] : R[0 + j];
            const float r1 = transpose ? R[3 * j + 1] : R[3 + j];
            const float r2 = transpose ? R[3 * j + 2] : R[6 + j];
            float v = p0 * r0;
            v = v + p1 * r1;
            v = v + p2 * r2;
            out[3 * i + j] = v;
        }
    }
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return WS_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        WS_HIP(hipMalloc((void**)&p, want * sizeof(T)));
        cap = want;
        return WS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// bucket counts of a default std::unordered_map as it grows: 13, 29, 59, 127, ...
const std::vector<unsigned long long>& prime_sequence()
{
    // built once, thread safe (the training thread and the prefetch thread may make their first call together):
    // initialisation of a function-local static is serialised by the language
    static const std::vector<unsigned long long> seq = [] {
        std::vector<unsigned long long> v;
        std::__detail::_Prime_rehash_policy pol;   // max_load_factor 1.0, like the reference's map
        size_t bkt = 1, n_elt = 0;
        while (v.size() < 40) {
            const auto r = pol._M_need_rehash(bkt, n_elt, 1);
            if (!r.first) break;                   // cannot happen for n_elt == bkt
            bkt = r.second;
            v.push_back((unsigned long long)bkt);
            n_elt = bkt;                           // next growth when the size reaches the bucket count
            if (bkt > (1ull << 33)) break;
        }
        return v;
    }();
    return seq;
}

}  // namespace

struct ws_subsample_ws {
    DevBuf<SubCloud> clouds;
    DevBuf<unsigned long long> tab_key, cell_key, primes;
    DevBuf<int32_t> tab_cnt, tab_first, slot_of, flag, first_slot, slot_rank, cell_cnt, cell_off, cursor, list,
        out_pos, scan_scratch, ord_scratch;
    std::vector<SubCloud> h_clouds;
    const float* points = nullptr;
    int64_t n = 0, m_total = 0, m_out = 0;
    int nb = 0, nprimes = 0;
    bool planned = false;
};

extern "C" {

int ws_subsample_ws_create(ws_subsample_ws** ws)
{
    WS_REQUIRE(ws, "NULL argument");
    *ws = new ws_subsample_ws();
    return WS_OK;
}

void ws_subsample_ws_destroy(ws_subsample_ws* w)
{
    if (!w) return;
    w->clouds.release(); w->tab_key.release(); w->cell_key.release(); w->primes.release();
    w->tab_cnt.release(); w->tab_first.release(); w->slot_of.release(); w->flag.release();
    w->first_slot.release(); w->slot_rank.release(); w->cell_cnt.release(); w->cell_off.release();
    w->cursor.release(); w->list.release(); w->out_pos.release(); w->scan_scratch.release();
    w->ord_scratch.release();
    delete w;
}

int ws_grid_subsample_plan(ws_subsample_ws* w, const float* points, int64_t n, const int32_t* h_lens, int32_t nb,
                           float dl, int32_t max_p, int32_t order_mode, int32_t* h_out_lens, int64_t* h_m, void* stream)
{
    WS_REQUIRE(w && h_lens && h_out_lens && h_m, "NULL argument");
    WS_REQUIRE(nb >= 1 && n >= 0, "bad sizes nb=%d n=%lld", nb, (long long)n);
    WS_REQUIRE(n < (1ll << 30), "point count exceeds int32 range");
    WS_REQUIRE(dl > 0.0f, "sampleDl must be > 0");
    WS_REQUIRE(order_mode == WS_ORDER_REFERENCE || order_mode == WS_ORDER_FIRST_SEEN, "unknown order_mode %d", order_mode);
    w->planned = false;
    *h_m = 0;
    hipStream_t st = (hipStream_t)stream;
    const auto& primes = prime_sequence();
    w->h_clouds.assign((size_t)nb, SubCloud{});
    int64_t sum = 0, tab = 0, ord = 0;
    for (int b = 0; b < nb; ++b) {
        WS_REQUIRE(h_lens[b] >= 0, "negative batch length");
        SubCloud& g = w->h_clouds[(size_t)b];
        g.base = (int)sum; g.len = h_lens[b];
        int64_t t = 64;
        while (t < 2 * (int64_t)h_lens[b]) t <<= 1;
        g.tab_base = (int)tab; g.tab_mask = (int)(t - 1);
        tab += t;
        g.org[0] = dl;   // parked for sub_bbox_kernel
        size_t pj = 0;
        while (pj + 1 < primes.size() && primes[pj] < (unsigned long long)h_lens[b]) ++pj;
        g.ord_bcap = (int)primes[pj];
        g.ord_base = (int)ord;
        ord += 5 * (int64_t)h_lens[b] + 3 * (int64_t)g.ord_bcap + 8;
        sum += h_lens[b];
        h_out_lens[b] = 0;
    }
    WS_REQUIRE(sum == n, "batch lengths sum to %lld, expected %lld", (long long)sum, (long long)n);
    WS_REQUIRE(tab < (1ll << 31) && ord < (1ll << 31), "workspace exceeds int32 indexing");
    if (n == 0) return ws_fail(WS_ERR_EMPTY, "Error");      // wrapper.cpp:266-270
    WS_REQUIRE(points, "NULL argument");

    int rc;
    if ((rc = w->clouds.ensure((size_t)nb))) return rc;
    if ((rc = w->tab_key.ensure((size_t)tab))) return rc;
    if ((rc = w->tab_cnt.ensure((size_t)tab))) return rc;
    if ((rc = w->tab_first.ensure((size_t)tab))) return rc;
    if ((rc = w->slot_rank.ensure((size_t)tab))) return rc;
    if ((rc = w->slot_of.ensure((size_t)n))) return rc;
    if ((rc = w->flag.ensure((size_t)n + 2))) return rc;
    if ((rc = w->first_slot.ensure((size_t)n))) return rc;
    if ((rc = w->cell_cnt.ensure((size_t)n + 2))) return rc;
    if ((rc = w->cell_off.ensure((size_t)n + 2))) return rc;
    if ((rc = w->cursor.ensure((size_t)n + 2))) return rc;
    if ((rc = w->cell_key.ensure((size_t)n))) return rc;
    if ((rc = w->list.ensure((size_t)n))) return rc;
    if ((rc = w->out_pos.ensure((size_t)n))) return rc;
    if ((rc = w->scan_scratch.ensure((size_t)ws_scan_scratch_items(n + 1)))) return rc;
    if ((rc = w->ord_scratch.ensure((size_t)ord))) return rc;
    if ((rc = w->primes.ensure(primes.size()))) return rc;

    WS_HIP(hipMemcpyAsync(w->clouds.p, w->h_clouds.data(), sizeof(SubCloud) * (size_t)nb, hipMemcpyHostToDevice, st));
    WS_HIP(hipMemcpyAsync(w->primes.p, primes.data(), sizeof(unsigned long long) * primes.size(), hipMemcpyHostToDevice, st));
    WS_HIP(hipMemsetAsync(w->tab_key.p, 0xff, sizeof(unsigned long long) * (size_t)tab, st));
    WS_HIP(hipMemsetAsync(w->tab_cnt.p, 0, sizeof(int32_t) * (size_t)tab, st));
    WS_HIP(hipMemsetAsync(w->tab_first.p, 0x7f, sizeof(int32_t) * (size_t)tab, st));
    WS_HIP(hipMemsetAsync(w->flag.p, 0, sizeof(int32_t) * (size_t)(n + 2), st));
    WS_HIP(hipMemsetAsync(w->cell_cnt.p, 0, sizeof(int32_t) * (size_t)(n + 2), st));

    sub_bbox_kernel<<<nb, 1024, 0, st>>>(points, w->clouds.p);
    WS_LAUNCH_CHECK();
    sub_insert_kernel<<<ws_grid(n, 256), 256, 0, st>>>(points, n, w->clouds.p, nb, dl, w->tab_key.p, w->tab_cnt.p,
                                                       w->tab_first.p, w->slot_of.p);
    WS_LAUNCH_CHECK();
    sub_mark_kernel<<<ws_grid(tab, 256), 256, 0, st>>>(tab, w->tab_cnt.p, w->tab_first.p, w->flag.p, w->first_slot.p);
    WS_LAUNCH_CHECK();
    // flag -> rank (exclusive scan, rank[n] = number of cells)
    if ((rc = ws_exclusive_scan_i32(w->flag.p, w->flag.p, n, w->scan_scratch.p, st))) return rc;
    sub_cells_kernel<<<ws_grid(n, 256), 256, 0, st>>>(n, w->flag.p, w->first_slot.p, w->tab_key.p, w->tab_cnt.p,
                                                      w->slot_rank.p, w->cell_cnt.p, w->cell_key.p);
    WS_LAUNCH_CHECK();
    sub_cloud_ranks_kernel<<<(nb + 63) / 64, 64, 0, st>>>(w->clouds.p, nb, w->flag.p, max_p);
    WS_LAUNCH_CHECK();
    // point lists of the cells
    if ((rc = ws_exclusive_scan_i32(w->cell_cnt.p, w->cell_off.p, n, w->scan_scratch.p, st))) return rc;
    WS_HIP(hipMemcpyAsync(w->cursor.p, w->cell_off.p, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, st));
    sub_list_fill_kernel<<<ws_grid(n, 256), 256, 0, st>>>(n, w->slot_of.p, w->slot_rank.p, w->cursor.p, w->list.p);
    WS_LAUNCH_CHECK();
    sub_list_sort_kernel<<<ws_grid(n, 256), 256, 0, st>>>(w->flag.p + n, w->cell_off.p, w->list.p);
    WS_LAUNCH_CHECK();
    sub_order_kernel<<<nb, 1024, 0, st>>>(w->clouds.p, w->cell_key.p, w->out_pos.p, w->ord_scratch.p, w->primes.p,
                                          (int)primes.size(), order_mode);
    WS_LAUNCH_CHECK();
    WS_HIP(hipMemcpyAsync(w->h_clouds.data(), w->clouds.p, sizeof(SubCloud) * (size_t)nb, hipMemcpyDeviceToHost, st));
    WS_HIP(hipStreamSynchronize(st));
    int64_t m_total = 0, m_out = 0;
    for (int b = 0; b < nb; ++b) {
        SubCloud& g = w->h_clouds[(size_t)b];
        g.out_base = (int)m_out;
        h_out_lens[b] = g.out_len;
        m_out += g.out_len;
        m_total += g.m;
    }
    WS_HIP(hipMemcpyAsync(w->clouds.p, w->h_clouds.data(), sizeof(SubCloud) * (size_t)nb, hipMemcpyHostToDevice, st));
    WS_HIP(hipStreamSynchronize(st));
    w->points = points; w->n = n; w->nb = nb; w->m_total = m_total; w->m_out = m_out;
    w->nprimes = (int)primes.size();
    *h_m = m_out;
    if (m_out == 0) return ws_fail(WS_ERR_EMPTY, "Error");
    w->planned = true;
    return WS_OK;
}

int ws_grid_subsample_fill(ws_subsample_ws* w, const float* features, int32_t fd, const int32_t* labels, int32_t ld,
                           float* out_points, float* out_features, int32_t* out_labels, uint64_t* out_keys,
                           int32_t* out_counts, void* stream)
{
    WS_REQUIRE(w && w->planned, "no successful plan to fill from");
    WS_REQUIRE(out_points, "out_points is NULL");
    WS_REQUIRE(!features || (fd >= 1 && out_features), "features given without fd/out_features");
    WS_REQUIRE(!labels || (ld >= 1 && out_labels), "labels given without ld/out_labels");
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)w->m_total;
    sub_emit_points_kernel<<<ws_grid(M, 256), 256, 0, st>>>(w->points, w->clouds.p, w->nb, M, w->cell_off.p, w->list.p,
                                                            w->out_pos.p, w->cell_key.p, out_points,
                                                            (unsigned long long*)out_keys, out_counts);
    WS_LAUNCH_CHECK();
    if (features) {
        sub_emit_features_kernel<<<ws_grid((int64_t)M * fd, 256), 256, 0, st>>>(features, fd, w->clouds.p, w->nb, M,
                                                                                w->cell_off.p, w->list.p, w->out_pos.p, out_features);
        WS_LAUNCH_CHECK();
    }
    if (labels) {
        sub_emit_labels_kernel<<<ws_grid((int64_t)M * ld, 64), 64, 0, st>>>(labels, ld, w->clouds.p, w->nb, M, w->cell_off.p,
                                                                            w->list.p, w->out_pos.p, w->primes.p, out_labels);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

int ws_rotate_clouds(const float* points, int64_t n, const int32_t* lens, int32_t nb, const float* rot,
                     int32_t transpose, float* out, void* stream)
{
    WS_REQUIRE(n >= 0 && nb >= 1, "bad sizes");
    if (n == 0) return WS_OK;
    WS_REQUIRE(points && lens && rot && out, "NULL argument");
    rotate_kernel<<<ws_grid(n, 256), 256, 0, (hipStream_t)stream>>>(points, n, lens, nb, rot, transpose, out);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_rotate_clouds_host(const float* points, int64_t n, const int32_t* h_lens, int32_t nb, const float* h_rot,
                          int32_t transpose, float* out, void* stream)
{
    WS_REQUIRE(n >= 0 && nb >= 1, "bad sizes");
    if (n == 0) return WS_OK;
    WS_REQUIRE(points && h_lens && h_rot && out, "NULL argument");
    if (nb > ROT_MAX) return ws_fail(WS_ERR_UNSUPPORTED, "more than %d batch elements: use ws_rotate_clouds", ROT_MAX);
    RotTable t;
    int64_t acc = 0;
    for (int b = 0; b < nb; ++b) {
        acc += h_lens[b];
        t.end[b] = (int32_t)acc;
        for (int e = 0; e < 9; ++e) t.r[9 * b + e] = h_rot[9 * b + e];
    }
    WS_REQUIRE(acc == n, "batch lengths sum to %lld, expected %lld", (long long)acc, (long long)n);
    rotate_table_kernel<<<ws_grid(n, 256), 256, 0, (hipStream_t)stream>>>(points, n, t, nb, transpose, out);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

const char* ws_last_error(void) { return ws_errbuf(); }
long long ws_launch_counter = 0;
int64_t ws_launch_count(void) { return (int64_t)__atomic_load_n(&ws_launch_counter, __ATOMIC_RELAXED); }
const char* ws_version(void) { return "weasal_hip 0.1 (gfx950)"; }
int ws_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
