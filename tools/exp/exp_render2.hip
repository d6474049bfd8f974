// Stand-alone A/B harness for the renderer kernels at BASELINE config-2 shapes (developer tool, not part of the product):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ispair_pytorch_amd/csrc [-DRF_TC=.. -DRF_ROWS=..] tools/exp/exp_render2.hip -o build/exp_r2
//   build/exp_r2 [object size as image fraction = 0.1875] [B = 256] [I = 128] [G = 16]
// Runs the first-generation forward / backward (render.hip) and the second-generation ones (render2.hip) on the same bf16
// sprites, prints the largest differences and the time per launch of each.
#include "../../spair_pytorch_amd/csrc/render.hip"
#include "../../spair_pytorch_amd/csrc/render2.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
template <class T> static T* dalloc(size_t n) {
    T* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); }
    hipMemset(p, 0, n * sizeof(T));
    return p;
}
static float maxdiff(const std::vector<float>& a, const std::vector<float>& b, float* ref = nullptr) {
    float m = 0.f, r = 0.f;
    for (size_t i = 0; i < a.size(); ++i) { m = fmaxf(m, fabsf(a[i] - b[i])); r = fmaxf(r, fabsf(b[i])); }
    if (ref) *ref = r;
    return m;
}

int main(int argc, char** argv) {
    const float size = argc > 1 ? atof(argv[1]) : 0.1875f;
    const int B = argc > 2 ? atoi(argv[2]) : 256, I = argc > 3 ? atoi(argv[3]) : 128, G = argc > 4 ? atoi(argv[4]) : 16;
    const int HW = G * G, P = 28, N = B * HW, PP2 = P * P * 2;
    std::vector<unsigned short> S((size_t)N * PP2);
    std::vector<float> nb((size_t)N * 4), pd((size_t)N * 2), x((size_t)B * I * I);
    srand(1);
    for (auto& v : S) { const _Float16 h = (_Float16)(rand() / (float)RAND_MAX); memcpy(&v, &h, 2); }     // fp16 (grey, alpha) pairs
    for (auto& v : x) v = (rand() % 4 == 0) ? rand() / (float)RAND_MAX : 0.f;
    for (int k = 0; k < HW; ++k) for (int b = 0; b < B; ++b) {
        const int r = k * B + b;
        nb[r * 4 + 0] = ((k % G) + rand() / (float)RAND_MAX) / G; nb[r * 4 + 1] = ((k / G) + rand() / (float)RAND_MAX) / G;
        nb[r * 4 + 2] = size * (0.8f + 0.4f * rand() / (float)RAND_MAX); nb[r * 4 + 3] = size * (0.8f + 0.4f * rand() / (float)RAND_MAX);
        pd[r * 2] = rand() / (float)RAND_MAX; pd[r * 2 + 1] = 4.f * rand() / (float)RAND_MAX;
    }
    unsigned short* dS = dalloc<unsigned short>(S.size());
    float *dnb = dalloc<float>(nb.size()), *dpd = dalloc<float>(pd.size()), *dx = dalloc<float>(x.size());
    hipMemcpy(dS, S.data(), S.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dnb, nb.data(), nb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpd, pd.data(), pd.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    const int nblk = render_num_blocks(B, I);
    float *rec[2], *aux[2], *part[2];
    for (int v = 0; v < 2; ++v) { rec[v] = dalloc<float>(x.size()); aux[v] = dalloc<float>(x.size() * 2); part[v] = dalloc<float>(nblk); }
    float* dgl = dalloc<float>(1);
    const float one = 1.f;
    hipMemcpy(dgl, &one, 4, hipMemcpyHostToDevice);
    unsigned short* dlog[2];
    float *dnbox[2], *dpres[2], *ddepth[2];
    for (int v = 0; v < 2; ++v) { dlog[v] = dalloc<unsigned short>((size_t)N * PP2); dnbox[v] = dalloc<float>((size_t)N * 4); dpres[v] = dalloc<float>(N); ddepth[v] = dalloc<float>(N); }
    const float* Sf = reinterpret_cast<const float*>(dS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto fwd = [&](int v) {
        if (v == 0) {
            hipLaunchKernelGGL(k_render_fwd<true>, dim3(nblk), dim3(256), 0, 0, Sf, PP2, dnb, dpd, dpd + 1, 2, dx, rec[0],
                               reinterpret_cast<float2*>(aux[0]), part[0], B, HW, I, P, 0);
            return 0;
        }
        return render_fwd2(Sf, PP2, dnb, dpd, dpd + 1, 2, dx, rec[1], aux[1], part[1], B, HW, I, P, 0, 1, 0);
    };
    auto time_it = [&](auto fn, const char* name, double bytes) {
        fn(); hipDeviceSynchronize();
        const int reps = 10;
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) fn();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us   %7.1f GB/s (algorithmic)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    };
    printf("B=%d I=%d G=%d object size %.3f (%.0f px)\n", B, I, G, size, size * I);
    const double fwd_bytes = (double)N * P * P * 2 * 2 + (double)N * 24 + (double)B * I * I * 4;
    int rc = fwd(1);
    if (rc != 0) printf("render_fwd2 rc = %d\n", rc);
    fwd(0);
    hipDeviceSynchronize();
    {
        std::vector<float> a(x.size()), b2(x.size()), ua(x.size() * 2), ub(x.size() * 2), pa(nblk), pb(nblk);
        hipMemcpy(a.data(), rec[0], a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b2.data(), rec[1], a.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ua.data(), aux[0], ua.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ub.data(), aux[1], ua.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(pa.data(), part[0], nblk * 4, hipMemcpyDeviceToHost); hipMemcpy(pb.data(), part[1], nblk * 4, hipMemcpyDeviceToHost);
        double sa = 0, sb = 0;
        for (int i = 0; i < nblk; ++i) { sa += pa[i]; sb += pb[i]; }
        std::vector<float> prea(x.size()), preb(x.size());
        for (size_t i = 0; i < x.size(); ++i) { prea[i] = ua[2 * i + 1]; preb[i] = ub[2 * i + 1]; }
        float r;
        printf("fwd v2 vs v1: max |recon diff| %.3g, max |pre diff| %.3g, BCE %.6g vs %.6g (rel %.2g)\n", maxdiff(b2, a, &r), maxdiff(preb, prea),
               sb, sa, fabs(sb - sa) / fabs(sa));
    }
    time_it([&] { fwd(0); }, "render fwd v1", fwd_bytes);
    time_it([&] { fwd(1); }, "render fwd v2", fwd_bytes);
#ifdef EXP_BWD2
    const double bwd_bytes = fwd_bytes + (double)N * P * P * 2 * 2;
    auto bwd = [&](int v) {
        if (v == 0) {
            const size_t lds = ((size_t)(P + 2) * (P + 2) * 4 + 3 * RB_CAP + I) * sizeof(float);
            hipLaunchKernelGGL(k_render_bwd<true>, dim3(B, HW), dim3(RB_T), lds, 0, Sf, PP2, dnb, dpd, dpd + 1, 2,
                               reinterpret_cast<const float2*>(aux[0]), dgl, reinterpret_cast<float*>(dlog[0]), dnbox[0], dpres[0], ddepth[0], PP2, B, HW, I,
                               P, 0, 2.0f, 0.1f, 1);
            return 0;
        }
        return render_bwd2(Sf, PP2, dnb, dpd, dpd + 1, 2, aux[0], dgl, reinterpret_cast<float*>(dlog[1]), dnbox[1], dpres[1], ddepth[1], PP2, B, HW, I, P, 0,
                           2.0f, 0.1f, 0);
    };
    rc = bwd(1);
    if (rc != 0) printf("render_bwd2 rc = %d\n", rc);
    bwd(0);
    hipDeviceSynchronize();
    {
        std::vector<unsigned short> la((size_t)N * PP2), lb((size_t)N * PP2);
        hipMemcpy(la.data(), dlog[0], la.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(lb.data(), dlog[1], la.size() * 2, hipMemcpyDeviceToHost);
        double num = 0, den = 0, dot = 0, nb2 = 0;
        float md = 0, mr = 0;
        for (size_t i = 0; i < la.size(); ++i) {
            const float a = bf2f(la[i]), b2 = bf2f(lb[i]);
            num += (double)(a - b2) * (a - b2); den += (double)a * a; dot += (double)a * b2; nb2 += (double)b2 * b2;
            md = fmaxf(md, fabsf(a - b2)); mr = fmaxf(mr, fabsf(a));
        }
        printf("bwd v2 vs v1: dlogits rel L2 %.3g, cos %.7f, max |diff| %.3g (max |ref| %.3g)\n", sqrt(num / den), dot / sqrt(den * nb2), md, mr);
        auto cmp = [&](float* p0, float* p1, size_t n, const char* nm) {
            std::vector<float> a(n), b2(n);
            hipMemcpy(a.data(), p0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b2.data(), p1, n * 4, hipMemcpyDeviceToHost);
            double nu = 0, de = 0;
            for (size_t i = 0; i < n; ++i) { nu += (double)(a[i] - b2[i]) * (a[i] - b2[i]); de += (double)a[i] * a[i]; }
            float r;
            const float m = maxdiff(b2, a, &r);
            printf("              %-8s rel L2 %.3g, max |diff| %.3g (max |ref| %.3g)\n", nm, sqrt(nu / de), m, r);
        };
        cmp(dnbox[0], dnbox[1], (size_t)N * 4, "dnbox"); cmp(dpres[0], dpres[1], N, "dpres"); cmp(ddepth[0], ddepth[1], N, "ddepth");
    }
    time_it([&] { bwd(0); }, "render bwd v1", bwd_bytes);
    time_it([&] { bwd(1); }, "render bwd v2", bwd_bytes);
#endif
    return 0;
}
