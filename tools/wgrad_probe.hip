// Stand-alone timing of csrc/wgrad.hip on the update's shapes, for trying main-loop variants:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Isafe-marl_amd/csrc ['-DWG_DEPTH(MT,NT)=3'] ['-DWG_UNROLL(MT,NT)=1'] \
//         tools/wgrad_probe.hip -o tools/wp/<name>      (tools/wp/ is git-ignored; the binaries travel to the GPU box)
#include "../safe-marl_amd/csrc/wgrad.hip"
#include <cstdio>
#include <cmath>
#include <vector>

int main() {
    struct Shape { int64_t k; int m, n; } shapes[] = {{163840, 192, 64}, {163840, 64, 144}, {163840, 4, 64}, {32768, 64, 720}, {32768, 64, 20}};
    float *a, *b, *c, *ws, *cs;
    const int64_t maxk = 163840;
    hipMalloc(&a, maxk * 192 * 4); hipMalloc(&b, maxk * 720 * 4); hipMalloc(&c, 192 * 745 * 4); hipMalloc(&cs, 192 * 4);
    hipMalloc(&ws, (int64_t)FLEXNET_WGRAD_WS_FLOATS * 4);
    {   // small integers: every product and partial sum is exact in fp32, so any two summation orders agree bit for bit
        std::vector<float> ha(maxk * 192), hb(maxk * 720);
        for (size_t i = 0; i < ha.size(); ++i) ha[i] = (float)((int)((i * 2654435761u) >> 29) - 3);
        for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((int)((i * 40503u + 7u) >> 5 & 7) - 3);
        hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    }
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto sh : shapes) {
        FlexWgradArgs w{};
        w.a = a; w.b = b; w.c = c; w.k = sh.k; w.m = sh.m; w.n = sh.n; w.lda = sh.m; w.ldb = sh.n; w.ldc = sh.n;
        w.workspace = ws; w.workspace_floats = FLEXNET_WGRAD_WS_FLOATS; w.accumulate = 0; w.colsum = cs;
        for (int i = 0; i < 5; ++i) if (flexnet_wgrad(&w, s) != 0) { printf("launch failed\n"); return 1; }
        const int reps = 50;
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) flexnet_wgrad(&w, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, gb = sh.k * (double)(sh.m + sh.n) * 4 / 1e9, tf = 2.0 * sh.k * sh.m * sh.n / 1e12;
        printf("k=%6ld m=%3d n=%3d  %7.1f us  %5.2f TB/s  %5.1f TFLOP/s", (long)sh.k, sh.m, sh.n, us, gb / us * 1e3, tf / us * 1e6);
        {   // checksum of C (compare between variants / WG_RING=0|1)
            std::vector<float> hc((size_t)sh.m * sh.n);
            hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
            double cs = 0, ca = 0; for (size_t i = 0; i < hc.size(); ++i) { cs += hc[i] * (double)((i % 97) + 1); ca += fabs(hc[i]); }
            printf("  checksum %.6e abs %.6e\n", cs, ca);
        }
    }
    return 0;
}
