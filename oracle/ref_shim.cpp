// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// extern "C" entry points around the *unmodified* reference C++ core, compiled
// from the sources where they lie under /root/reference by oracle/Makefile
// (target `ref`).  Output: oracle/_ref/libws_ref.so (git-ignored, built here,
// travels to the GPU box as a prebuilt file).
//
//   ref_batch_query      -> batch_nanoflann_neighbors   cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332
//   ref_subsample_batch  -> batch_grid_subsampling      cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
//   ref_subsample        -> grid_subsampling            cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
//
// The shim only converts raw pointers to the std::vector arguments the
// reference functions take (the job of the CPython glue in wrapper.cpp, which
// does not compile against NumPy 2.x) and mallocs the results.
#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"
#include <cstdlib>
#include <cstring>

extern "C" {

// returns 0 on success; *out is malloc'd [nq * *max_count] int32 (free with ref_free)
int ref_batch_query(const float* q, int nq, const float* s, int ns,
                    const int* qb, const int* sb, int nb, float radius,
                    int** out, int* max_count)
{
    vector<PointXYZ> queries((const PointXYZ*)q, (const PointXYZ*)q + nq);
    vector<PointXYZ> supports((const PointXYZ*)s, (const PointXYZ*)s + ns);
    vector<int> q_batches(qb, qb + nb), s_batches(sb, sb + nb);
    vector<int> ind;
    batch_nanoflann_neighbors(queries, supports, q_batches, s_batches, ind, radius);
    if (ind.size() < 1) { *out = nullptr; *max_count = 0; return 1; }
    *max_count = (int)(ind.size() / (size_t)nq);
    *out = (int*)malloc(ind.size() * sizeof(int));
    memcpy(*out, ind.data(), ind.size() * sizeof(int));
    return 0;
}

int ref_subsample_batch(const float* p, int n, const int* lens, int nb,
                        const float* feat, int fd, const int* cls, int ld,
                        float dl, int max_p,
                        float** out_p, int* out_lens, float** out_f, int** out_c, int* m)
{
    vector<PointXYZ> pts((const PointXYZ*)p, (const PointXYZ*)p + n);
    vector<int> batches(lens, lens + nb);
    vector<float> f; if (feat) f.assign(feat, feat + (size_t)n * fd);
    vector<int> c;   if (cls)  c.assign(cls, cls + (size_t)n * ld);
    vector<PointXYZ> sp; vector<float> sf; vector<int> sc; vector<int> sb;
    batch_grid_subsampling(pts, sp, f, sf, c, sc, batches, sb, dl, max_p);
    *m = (int)sp.size();
    if (sp.size() < 1) return 1;
    *out_p = (float*)malloc(sp.size() * 3 * sizeof(float));
    memcpy(*out_p, sp.data(), sp.size() * 3 * sizeof(float));
    memcpy(out_lens, sb.data(), nb * sizeof(int));
    if (feat) { *out_f = (float*)malloc(sf.size() * sizeof(float)); memcpy(*out_f, sf.data(), sf.size() * sizeof(float)); }
    if (cls)  { *out_c = (int*)malloc(sc.size() * sizeof(int));     memcpy(*out_c, sc.data(), sc.size() * sizeof(int)); }
    return 0;
}

int ref_subsample(const float* p, int n, const float* feat, int fd, const int* cls, int ld,
                  float dl, float** out_p, float** out_f, int** out_c, int* m)
{
    vector<PointXYZ> pts((const PointXYZ*)p, (const PointXYZ*)p + n);
    vector<float> f; if (feat) f.assign(feat, feat + (size_t)n * fd);
    vector<int> c;   if (cls)  c.assign(cls, cls + (size_t)n * ld);
    vector<PointXYZ> sp; vector<float> sf; vector<int> sc;
    grid_subsampling(pts, sp, f, sf, c, sc, dl, 0);
    *m = (int)sp.size();
    if (sp.size() < 1) return 1;
    *out_p = (float*)malloc(sp.size() * 3 * sizeof(float));
    memcpy(*out_p, sp.data(), sp.size() * 3 * sizeof(float));
    if (feat) { *out_f = (float*)malloc(sf.size() * sizeof(float)); memcpy(*out_f, sf.data(), sf.size() * sizeof(float)); }
    if (cls)  { *out_c = (int*)malloc(sc.size() * sizeof(int));     memcpy(*out_c, sc.data(), sc.size() * sizeof(int)); }
    return 0;
}

void ref_free(void* p) { free(p); }

}  // extern "C"
