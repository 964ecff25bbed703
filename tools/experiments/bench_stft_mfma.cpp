// Phase timing of the matrix-core STFT kernel (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value -ffp-contract=on \
//         -DDSA_STFT_TIMING tools/bench_stft_mfma.cpp -o build/bench_stft_mfma && build/bench_stft_mfma [B]
#include "../diffsptk_amd/csrc/stft.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char** argv)
{
    long B = argc > 1 ? atol(argv[1]) : 1024, T = 16000;
    const int L = 400, P = 80;
    long N = (T - 1) / P + 1;
    std::vector<float> hx(B * T), hw(L);
    srand(1);
    for (auto& v : hx) v = (float)(rand() % 65536) / 32768.f - 1.f;
    for (int i = 0; i < L; ++i) hw[i] = 0.05f * (0.42f - 0.5f * cosf(6.2831853f * i / (L - 1)) + 0.08f * cosf(12.566371f * i / (L - 1)));
    float *x, *w, *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, L * 4); hipMalloc(&y, B * N * 257 * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), L * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 23; ++rep) {
        float ms;
        hipEventRecord(e0);
        dsa::stft512_mfma_launch(x, B, T, N, L, P, 200, 0, w, 1e-9f, 3, y, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_stft_stamps), sizeof(st));
    printf("B=%ld: min %.1f us median %.1f us (%.0f GB/s) | third tile of wave 0, cycles: load+amax %llu  reload+split %llu  mfma+combine %llu  store %llu\n",
           B, ts[0], ts[ts.size() / 2], 1348.0 * B * N / ts[ts.size() / 2] / 1e3, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]);
    return 0;
}
