// Kernel-only timing of the tuned mcep forward for compile-time A/B experiments (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value [-DDSA_...] \
//         tools/ab_kernel.cpp -o build/ab_<tag> && build/ab_<tag> [frames] [reps]
#include "../diffsptk_amd/csrc/mcep_mfma.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char** argv)
{
    long F = argc > 1 ? atol(argv[1]) : 204800;
    int reps = argc > 2 ? atoi(argv[2]) : 30;
    const int K = 257, M1 = 25, M2 = 49;
    std::vector<float> hX(F * K), hG(K * M1), hD(M1 * K), hE(K * M2), hav(M1);
    srand(1);
    for (auto& v : hX) v = 0.5f + (float)(rand() % 65536) / 65536.f;
    for (int k = 0; k < K; ++k)
        for (int m = 0; m < M1; ++m) {
            hG[k * M1 + m] = (m == 0 ? 1.f : 0.1f) / K * cosf(0.01f * k * m);
            hD[m * K + k] = cosf(0.012f * k * m) * (m == 0 ? 1.f : 0.05f);
        }
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < M2; ++j) hE[k * M2 + j] = cosf(3.14159265f * k * j / 256.f) / 512.f * (k == 0 || k == 256 ? 1.f : 2.f);
    for (int m = 0; m < M1; ++m) hav[m] = powf(-0.42f, (float)m);
    float *X, *G, *D, *E, *av, *mc;
    hipMalloc(&X, hX.size() * 4); hipMalloc(&G, hG.size() * 4); hipMalloc(&D, hD.size() * 4);
    hipMalloc(&E, hE.size() * 4); hipMalloc(&av, 100); hipMalloc(&mc, F * M1 * 4);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(av, hav.data(), 100, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < reps + 3; ++rep) {
        float ms = 0;
        hipEventRecord(e0);
        dsa::launch_h<8>(X, F, 10, G, D, E, av, mc, nullptr, 0, "h8");
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    std::vector<float> h(4);
    hipMemcpy(h.data(), mc, 16, hipMemcpyDeviceToHost);
    printf("%s: min %.4f  median %.4f  max %.4f ms  (mc %g %g)\n", argv[0], ts[0], ts[ts.size() / 2], ts.back(), h[0], h[1]);
    return 0;
}
