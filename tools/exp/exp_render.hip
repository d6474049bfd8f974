// Stand-alone timing harness for k_render_bwd at BASELINE config-2 shapes (developer tool, not part of the product):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ispair_pytorch_amd/csrc [-DRB_WAVES=n -DRB_CAP=n] tools/exp/exp_render.hip -o build/exp_rb
#ifndef RB_EXP
#define RB_EXP 0
#endif
#include "../../spair_pytorch_amd/csrc/render.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
int main(int argc, char** argv) {
    const int B = 256, G = 16, HW = G * G, I = 128, P = 28;
    const float size = argc > 1 ? atof(argv[1]) : 0.1875f;
    const int N = B * HW, PP2 = P * P * 2;
    std::vector<float> S((size_t)N * PP2), nb((size_t)N * 4), pd((size_t)N * 2), aux((size_t)B * I * I * 4);
    srand(1);
    for (auto& v : S) v = rand() / (float)RAND_MAX;
    for (auto& v : aux) v = rand() / (float)RAND_MAX + 0.5f;
    for (int k = 0; k < HW; ++k) for (int b = 0; b < B; ++b) {
        const int r = k * B + b;
        nb[r * 4 + 0] = ((k % G) + rand() / (float)RAND_MAX) / G; nb[r * 4 + 1] = ((k / G) + rand() / (float)RAND_MAX) / G;
        nb[r * 4 + 2] = size * (0.8f + 0.4f * rand() / (float)RAND_MAX); nb[r * 4 + 3] = size * (0.8f + 0.4f * rand() / (float)RAND_MAX);
        pd[r * 2] = 0.5f; pd[r * 2 + 1] = 0.5f;
    }
    float *dS, *dnb, *dpd, *daux, *dgl, *dlog, *dnbox, *dpres, *ddepth;
    hipMalloc(&dS, S.size() * 4); hipMalloc(&dnb, nb.size() * 4); hipMalloc(&dpd, pd.size() * 4); hipMalloc(&daux, aux.size() * 4);
    hipMalloc(&dgl, 4); hipMalloc(&dlog, (size_t)N * PP2 * 4); hipMalloc(&dnbox, (size_t)N * 16); hipMalloc(&dpres, (size_t)N * 4); hipMalloc(&ddepth, (size_t)N * 4);
    hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dnb, nb.data(), nb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpd, pd.data(), pd.size() * 4, hipMemcpyHostToDevice); hipMemcpy(daux, aux.data(), aux.size() * 4, hipMemcpyHostToDevice);
    float one = 1.f; hipMemcpy(dgl, &one, 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i)
            render_bwd(dS, PP2, dnb, dpd, dpd + 1, 2, daux, dgl, dlog, dnbox, dpres, ddepth, PP2, B, HW, 1, I, P, 0, 1.f, 1.f, 1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("RB_EXP=%d size=%.3f: %.3f ms\n", RB_EXP, size, ms / 5);
    }
    std::vector<float> o(8); hipMemcpy(o.data(), dnbox, 32, hipMemcpyDeviceToHost);
    printf("  dnbox[0..3] = %g %g %g %g\n", o[0], o[1], o[2], o[3]);
    return 0;
}
