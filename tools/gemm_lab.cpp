// tools/gemm_lab.cpp -- diagnostic: time the tall-skinny GEMM variants of libweasal_hip.so per shape.
// build: hipcc --offload-arch=gfx950 -O2 -Iinclude tools/gemm_lab.cpp -Lweasal_amd -lweasal_hip -Wl,-rpath,$PWD/weasal_amd -o gpurun_out/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "weasal_hip.h"
extern "C" int ws_gemm_variant;
extern "C" int ws_gemm_wave_cols;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Shape { int64_t m; int k, n, ep; };
int main(int argc, char** argv)
{
    std::vector<Shape> shapes = {
        {400000, 128, 128, 0}, {400000, 128, 128, 6}, {400000, 32, 128, 6}, {400000, 32, 480, 0}, {400000, 480, 32, 4},
        {400000, 64, 128, 0}, {400000, 128, 64, 0}, {400000, 128, 32, 0}, {400000, 32, 64, 0}, {400000, 64, 32, 4},
        {71070, 256, 256, 6}, {71070, 960, 64, 4}, {71070, 64, 960, 0}, {71070, 256, 128, 0}, {71070, 128, 256, 0},
        {10257, 1920, 128, 4}, {10257, 512, 512, 6}, {10257, 512, 256, 0}, {10257, 128, 1920, 0}, {10257, 960, 64, 4}, {10257, 256, 64, 0}};
    float *x, *b, *y1, *y2, *res, *bias;
    const size_t maxx = 400000ull * 512, maxy = 400000ull * 512;
    CK(hipMalloc(&x, maxx * 4)); CK(hipMalloc(&y1, maxy * 4)); CK(hipMalloc(&y2, maxy * 4)); CK(hipMalloc(&res, maxy * 4));
    CK(hipMalloc(&b, 1920 * 1920 * 4)); CK(hipMalloc(&bias, 4096 * 4));
    std::vector<float> h(maxx);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, h.data(), maxx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, h.data(), maxy * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), 1920 * 1920 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data() + 77, 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> o1, o2;
    const int first = argc > 1 ? atoi(argv[1]) : 0, count = argc > 2 ? atoi(argv[2]) : (int)shapes.size();
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const int ablate = argc > 4 ? atoi(argv[4]) : 0;
    for (int si = first; first >= 0 && si < first + count && si < (int)shapes.size(); ++si) {
        const Shape s = shapes[si];
        float tm[5] = {0, 0, 0, 0, 0};
        for (int var = 1; var <= 3; ++var) {
            ws_gemm_variant = var >= 2 ? 2 : 1;
            ws_gemm_wave_cols = var == 2 ? 1 : (var == 3 ? 2 : 0);
            float* y = var == 1 ? y1 : y2;
            auto run = [&]() {
                int rc = ws_gemm_xb_epilogue(x, s.m, s.k, s.k, b, s.n, (s.ep & 1) ? bias : nullptr, (s.ep & 2) ? res : nullptr, s.n,
                                             ((s.ep & 4) ? 1 : 0) , 0.1f, y, s.n, nullptr);
                if (rc) { printf("error: %s\n", ws_last_error()); exit(1); }
            };
            for (int i = 0; i < 3; ++i) run();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) run();
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&tm[var], e0, e1)); tm[var] *= 1000.f / reps;
        }
        const size_t cnt = (size_t)s.m * s.n;
        o1.resize(cnt); o2.resize(cnt);
        CK(hipMemcpy(o1.data(), y1, cnt * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), y2, cnt * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < cnt; ++i) { md = fmax(md, fabs((double)o1[i] - o2[i])); mx = fmax(mx, fabs((double)o1[i])); }
        const double byt = 4.0 * s.m * (s.k + s.n) + ((s.ep & 2) ? 4.0 * s.m * s.n : 0), flop = 2.0 * s.m * s.k * s.n;
        const double floor_us = fmax(byt / 5.0e12, flop / 150e12) * 1e6;
        printf("M=%7lld K=%4d N=%4d ep=%d  v1 %7.1f us  wn1 %7.1f  wn2 %7.1f  wn4 %7.1f  floor %6.1f  eff %.2f -> %.2f -> %.2f  maxdiff %.2e (max %.1f)\n",
               (long long)s.m, s.k, s.n, s.ep, tm[1], tm[2], tm[3], tm[4], floor_us, floor_us / tm[1], floor_us / tm[2], floor_us / tm[3], md, mx);
        fflush(stdout);
    }
    // ---- X^T Y reductions
    std::vector<Shape> tshapes = {{400000, 128, 128, 0}, {400000, 480, 32, 0}, {400000, 64, 128, 0}, {400000, 128, 32, 0}, {400000, 32, 128, 0},
        {400000, 64, 32, 0}, {400000, 128, 9, 0}, {400000, 45, 64, 0}, {71070, 960, 64, 0}, {71070, 256, 256, 0}, {71070, 128, 256, 0}, {71070, 480, 32, 0},
        {10257, 1920, 128, 0}, {10257, 512, 512, 0}, {10257, 128, 512, 0}, {10257, 64, 256, 0}};
    void* scratch; CK(hipMalloc(&scratch, 512ull << 20));
    float *oa, *ob; CK(hipMalloc(&oa, 1920 * 1920 * 4)); CK(hipMalloc(&ob, 1920 * 1920 * 4));
    if (argc <= 1 || atoi(argv[1]) < 0)
    for (auto s : tshapes) {
        float tm[3] = {0, 0, 0};
        for (int var = 1; var <= 2; ++var) {
            ws_gemm_variant = var;
            float* o = var == 1 ? oa : ob;
            auto run = [&]() {
                int rc = ws_gemm_xty(x, s.m, s.k, s.k, res, s.n, s.n, o, scratch, nullptr);
                if (rc) { printf("error: %s\n", ws_last_error()); exit(1); }
            };
            for (int i = 0; i < 3; ++i) run();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) run();
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&tm[var], e0, e1)); tm[var] *= 1000.f / 20;
        }
        const size_t cnt = (size_t)s.k * s.n;
        o1.resize(cnt); o2.resize(cnt);
        CK(hipMemcpy(o1.data(), oa, cnt * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), ob, cnt * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < cnt; ++i) { md = fmax(md, fabs((double)o1[i] - o2[i])); mx = fmax(mx, fabs((double)o1[i])); }
        const double byt = 4.0 * s.m * (s.k + s.n), flop = 2.0 * s.m * s.k * s.n;
        const double floor_us = fmax(byt / 5.0e12, flop / 150e12) * 1e6;
        printf("XTY M=%7lld K=%4d N=%4d  v1 %7.1f us  v2 %7.1f us  floor %6.1f  eff %.2f -> %.2f   maxdiff %.2e (max %.1f)\n",
               (long long)s.m, s.k, s.n, tm[1], tm[2], floor_us, floor_us / tm[1], floor_us / tm[2], md, mx);
        fflush(stdout);
    }
    return 0;
}
