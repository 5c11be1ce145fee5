// oracle/ws_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's native geometry (radius neighbours and
// grid subsampling), written from the arithmetic contracts in SURVEY.md
// Appendix A.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load the library built from this file
// (oracle/libws_oracle.so); the product (weasal_amd/) never does.
//
// Pinning: tests/test_oracle_cpu.py checks every function below against
//   (1) the golden vectors in tests/golden/ (generated from the reference's own
//       compiled core, see tests/golden/make_golden.py), and
//   (2) oracle/_ref/libws_ref.so (the unmodified reference sources compiled by
//       oracle/Makefile) whenever that file is present.
//
// Build: g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared (oracle/Makefile).
// -ffp-contract=off: the reference objects contain no FMA; every f32 product
// and sum below must round separately.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct P3 { float x, y, z; };  // reference: PointXYZ, cpp_utils/cloud/cloud.h:40-104

// d2 exactly as nanoflann's L2_Simple_Adaptor evaluates it
// (cpp_utils/nanoflann/nanoflann.hpp:432-440): result = 0; result += diff*diff per dim.
inline float sqdist(const P3& a, const P3& b)
{
    float r = 0.0f;
    float d = a.x - b.x; r += d * d;
    d = a.y - b.y;       r += d * d;
    d = a.z - b.z;       r += d * d;
    return r;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------
// Radius neighbours.  Follows batch_nanoflann_neighbors,
// cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332:
//   r2 = radius*radius (f32, :226); per batch element the supports slice is
//   searched for every query of the same element (:272-293); matches are
//   d2 < r2 (nanoflann.hpp:249-253 RadiusResultSet::addPoint, strict), sorted
//   ascending by d2 (nanoflann.hpp:1287, IndexDist_Sorter :208-214 -- ties
//   unspecified there, broken by index here); rows are padded to the global
//   max count with supports.size() and local indices are offset by the sum of
//   the previous support lengths (:316-326).
// The kd-tree is replaced by a uniform grid (cell = radius) -- the result set
// is defined by the distance predicate, not by the search structure.
// Returns 0 on success, 1 when the result is empty (wrapper.cpp:201-205).
// ---------------------------------------------------------------------------
int orc_radius_neighbors(const float* q_, int nq, const float* s_, int ns,
                         const int* qb, const int* sb, int nb, float radius,
                         int** out, int* max_count)
{
    const P3* q = (const P3*)q_;
    const P3* s = (const P3*)s_;
    const float r2 = radius * radius;
    std::vector<std::vector<std::pair<float, int>>> rows((size_t)nq);
    int maxc = 0;
    int q0 = 0, s0 = 0;
    for (int b = 0; b < nb; ++b) {
        const int nqb = qb[b], nsb = sb[b];
        if (nsb > 0 && nqb > 0) {
            // bounding box of the supports of this element
            P3 lo = s[s0], hi = s[s0];
            for (int j = 1; j < nsb; ++j) {
                const P3& p = s[s0 + j];
                lo.x = std::min(lo.x, p.x); lo.y = std::min(lo.y, p.y); lo.z = std::min(lo.z, p.z);
                hi.x = std::max(hi.x, p.x); hi.y = std::max(hi.y, p.y); hi.z = std::max(hi.z, p.z);
            }
            // cell size >= radius, grid capped at 128^3 cells
            double cell = radius > 0 ? (double)radius : 1.0;
            double ext = std::max({(double)hi.x - lo.x, (double)hi.y - lo.y, (double)hi.z - lo.z});
            if (ext / cell > 127.0) cell = ext / 127.0;
            auto cidx = [&](double v, double l) { return (int)std::floor((v - l) / cell); };
            const int nx = cidx(hi.x, lo.x) + 1, ny = cidx(hi.y, lo.y) + 1, nz = cidx(hi.z, lo.z) + 1;
            std::vector<int> start((size_t)nx * ny * nz + 1, 0);
            std::vector<int> cell_of((size_t)nsb);
            for (int j = 0; j < nsb; ++j) {
                const P3& p = s[s0 + j];
                int c = (cidx(p.z, lo.z) * ny + cidx(p.y, lo.y)) * nx + cidx(p.x, lo.x);
                cell_of[j] = c;
                start[(size_t)c + 1]++;
            }
            for (size_t c = 0; c + 1 < start.size(); ++c) start[c + 1] += start[c];
            std::vector<int> order((size_t)nsb), cur(start.begin(), start.end() - 1);
            for (int j = 0; j < nsb; ++j) order[(size_t)cur[cell_of[j]]++] = j;

            for (int i = 0; i < nqb; ++i) {
                const P3& p = q[q0 + i];
                auto& row = rows[(size_t)q0 + i];
                const int cx = cidx(p.x, lo.x), cy = cidx(p.y, lo.y), cz = cidx(p.z, lo.z);
                for (int dz = -1; dz <= 1; ++dz) {
                    int z = cz + dz; if (z < 0 || z >= nz) continue;
                    for (int dy = -1; dy <= 1; ++dy) {
                        int y = cy + dy; if (y < 0 || y >= ny) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            int x = cx + dx; if (x < 0 || x >= nx) continue;
                            size_t c = ((size_t)z * ny + y) * nx + x;
                            for (int k = start[c]; k < start[c + 1]; ++k) {
                                int j = order[(size_t)k];
                                float d2 = sqdist(p, s[s0 + j]);
                                if (d2 < r2) row.emplace_back(d2, j + s0);
                            }
                        }
                    }
                }
                std::sort(row.begin(), row.end());
                maxc = std::max(maxc, (int)row.size());
            }
        }
        q0 += nqb; s0 += nsb;
    }
    *max_count = maxc;
    if (nq == 0 || maxc == 0) { *out = nullptr; return 1; }
    int* o = (int*)malloc((size_t)nq * maxc * sizeof(int));
    for (int i = 0; i < nq; ++i) {
        const auto& row = rows[(size_t)i];
        for (int j = 0; j < maxc; ++j)
            o[(size_t)i * maxc + j] = j < (int)row.size() ? row[(size_t)j].second : ns;
    }
    *out = o;
    return 0;
}

// ---------------------------------------------------------------------------
// Grid subsampling.  Follows grid_subsampling / batch_grid_subsampling,
// cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106 and
// :109-211, SampledData grid_subsampling.h:10-80, min_point/max_point
// cpp_utils/cloud/cloud.cpp:27-66.  All arithmetic f32 except the reciprocal
// of the count (double, narrowed to f32 when it meets PointXYZ::operator*,
// grid_subsampling.cpp:87).  Row order = iteration order of a libstdc++
// std::unordered_map<size_t, .> filled in first-occurrence order (:48,:85).
// Also emits per output row the cell key and point count (test aids).
// Label = first max of an unordered_map<int,int> histogram (:99-101).
// ---------------------------------------------------------------------------
static void subsample_one(const P3* p, int n, const float* feat, int fd, const int* cls, int ld,
                          float dl, std::vector<P3>& op, std::vector<float>& of, std::vector<int>& oc,
                          std::vector<uint64_t>& okey, std::vector<int>& ocnt)
{
    if (n <= 0) return;
    P3 mn = p[0], mx = p[0];
    for (int i = 0; i < n; ++i) {
        if (p[i].x < mn.x) mn.x = p[i].x;
        if (p[i].y < mn.y) mn.y = p[i].y;
        if (p[i].z < mn.z) mn.z = p[i].z;
        if (p[i].x > mx.x) mx.x = p[i].x;
        if (p[i].y > mx.y) mx.y = p[i].y;
        if (p[i].z > mx.z) mx.z = p[i].z;
    }
    const float inv = 1 / dl;                                  // (1/sampleDl), :27
    P3 org;
    org.x = std::floor(mn.x * inv) * dl;
    org.y = std::floor(mn.y * inv) * dl;
    org.z = std::floor(mn.z * inv) * dl;
    const size_t nX = (size_t)std::floor((mx.x - org.x) / dl) + 1;   // :30
    const size_t nY = (size_t)std::floor((mx.y - org.y) / dl) + 1;   // :31

    struct Cell { int count; P3 sum; std::vector<float> f; std::vector<std::unordered_map<int, int>> lab; };
    std::unordered_map<size_t, Cell> data;
    for (int i = 0; i < n; ++i) {
        size_t iX = (size_t)std::floor((p[i].x - org.x) / dl);
        size_t iY = (size_t)std::floor((p[i].y - org.y) / dl);
        size_t iZ = (size_t)std::floor((p[i].z - org.z) / dl);
        size_t key = iX + nX * iY + nX * nY * iZ;
        if (data.count(key) < 1) {
            Cell c; c.count = 0; c.sum = {0, 0, 0};
            c.f.assign((size_t)fd, 0.0f); c.lab.resize((size_t)ld);
            data.emplace(key, c);
        }
        Cell& c = data[key];
        c.count += 1;
        c.sum.x += p[i].x; c.sum.y += p[i].y; c.sum.z += p[i].z;
        for (int k = 0; k < fd; ++k) c.f[(size_t)k] += feat[(size_t)i * fd + k];
        for (int k = 0; k < ld; ++k) c.lab[(size_t)k][cls[(size_t)i * ld + k]] += 1;
    }
    for (auto& kv : data) {
        Cell& c = kv.second;
        const float a = (float)(1.0 / c.count);
        op.push_back({c.sum.x * a, c.sum.y * a, c.sum.z * a});
        okey.push_back((uint64_t)kv.first);
        ocnt.push_back(c.count);
        const float cnt = (float)c.count;
        for (int k = 0; k < fd; ++k) of.push_back(c.f[(size_t)k] / cnt);       // :90-94
        for (int k = 0; k < ld; ++k) {
            auto best = std::max_element(c.lab[(size_t)k].begin(), c.lab[(size_t)k].end(),
                [](const std::pair<const int, int>& x, const std::pair<const int, int>& y) { return x.second < y.second; });
            oc.push_back(best->first);
        }
    }
}

// feat/cls may be NULL (fd/ld then ignored).  lens==NULL -> single cloud (nb ignored).
// out_key/out_cnt may be NULL.  Returns 0, or 1 when the result is empty.
int orc_grid_subsample_batch(const float* p_, int n, const int* lens, int nb,
                             const float* feat, int fd, const int* cls, int ld,
                             float dl, int max_p,
                             float** out_p, int* out_lens, float** out_f, int** out_c,
                             uint64_t** out_key, int** out_cnt, int* m)
{
    const P3* p = (const P3*)p_;
    if (!feat) fd = 0;
    if (!cls) ld = 0;
    std::vector<P3> op; std::vector<float> of; std::vector<int> oc;
    std::vector<uint64_t> okey; std::vector<int> ocnt;
    int one = n;
    if (!lens) { lens = &one; nb = 1; }
    if (max_p < 1) max_p = n;                                      // :133-134
    int s0 = 0;
    for (int b = 0; b < nb; ++b) {
        std::vector<P3> bp; std::vector<float> bf; std::vector<int> bc;
        std::vector<uint64_t> bk; std::vector<int> bn;
        subsample_one(p + s0, lens[b], feat ? feat + (size_t)s0 * fd : nullptr, fd,
                      cls ? cls + (size_t)s0 * ld : nullptr, ld, dl, bp, bf, bc, bk, bn);
        size_t keep = std::min(bp.size(), (size_t)max_p);           // :181-204
        op.insert(op.end(), bp.begin(), bp.begin() + keep);
        of.insert(of.end(), bf.begin(), bf.begin() + keep * fd);
        oc.insert(oc.end(), bc.begin(), bc.begin() + keep * ld);
        okey.insert(okey.end(), bk.begin(), bk.begin() + keep);
        ocnt.insert(ocnt.end(), bn.begin(), bn.begin() + keep);
        if (out_lens) out_lens[b] = (int)keep;
        s0 += lens[b];
    }
    *m = (int)op.size();
    if (op.empty()) return 1;
    auto dup = [](const void* src, size_t bytes) { void* d = malloc(bytes ? bytes : 1); memcpy(d, src, bytes); return d; };
    *out_p = (float*)dup(op.data(), op.size() * sizeof(P3));
    if (out_f && fd) *out_f = (float*)dup(of.data(), of.size() * sizeof(float));
    if (out_c && ld) *out_c = (int*)dup(oc.data(), oc.size() * sizeof(int));
    if (out_key) *out_key = (uint64_t*)dup(okey.data(), okey.size() * sizeof(uint64_t));
    if (out_cnt) *out_cnt = (int*)dup(ocnt.data(), ocnt.size() * sizeof(int));
    return 0;
}

void orc_free(void* p) { free(p); }

}  // extern "C"
