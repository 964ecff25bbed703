// Phase timing of the split-precision mcep backward kernel (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value -ffp-contract=on \
//         -DDSA_MCEP_TIMING tools/bench_mcep_bwd.cpp -o build/bench_mcep_bwd && build/bench_mcep_bwd [frames]
#include "../diffsptk_amd/csrc/mcep_mfma.hip"

#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char** argv)
{
    long F = argc > 1 ? atol(argv[1]) : 51200;
    const int K = 257, M1 = 25, M2 = 49, NI = 10;
    std::vector<float> hX(F * K), hG(K * M1), hD(M1 * K), hE(K * M2), hav(M1), hg(F * M1, 0.01f);
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
    float *X, *G, *D, *E, *av, *mc, *hist, *gm, *gX;
    hipMalloc(&X, hX.size() * 4); hipMalloc(&G, hG.size() * 4); hipMalloc(&D, hD.size() * 4); hipMalloc(&E, hE.size() * 4);
    hipMalloc(&av, 100); hipMalloc(&mc, F * M1 * 4); hipMalloc(&hist, (size_t)(NI + 1) * F * M1 * 4); hipMalloc(&gm, F * M1 * 4);
    hipMalloc(&gX, hX.size() * 4);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(av, hav.data(), 100, hipMemcpyHostToDevice);
    hipMemcpy(gm, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    dsa::launch_h<8>(X, F, NI, G, D, E, av, mc, hist, 0, "h8");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        dsa::mcep_mfma_bwd_h(gm, X, hist, F, NI, G, D, E, av, gX, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_mcep_stamps), sizeof(st));
    printf("F=%ld bwd %.3f ms | step cycles: hist+fwd chains %llu  solve(2 rhs) %llu  rtbar+exchange %llu  ebar chain %llu  mbar chain %llu | whole step %llu\n",
           F, ms, st[41] - st[40], st[42] - st[41], st[43] - st[42], st[44] - st[43], st[45] - st[44], st[45] - st[40]);
    return 0;
}
