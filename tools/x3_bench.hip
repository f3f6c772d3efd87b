// x3_bench.hip -- pdgn_gemm_nt (gemm_x3.hip) alone, with compile-time ablations (-DX3_ABLATE=n) to see where its time goes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 -DX3_ABLATE=0 [-DX3_DEFAULT_SHAPE=16] \
//         -Ipdgn_amd/csrc tools/x3_bench.hip pdgn_amd/csrc/gemm_x3_16.hip pdgn_amd/csrc/gemm_x3_h2.hip -o /tmp/x3b
// PDGN_GEMM=x3: three bf16 parts everywhere; default: two fp16 parts where they pay (the harness provides the scale slots)
#ifdef X3_DEFAULT_SHAPE                 // -DX3_DEFAULT_SHAPE=16: every instance class on v_mfma_f32_16x16x32_bf16
#define X3_DEFAULT_MASK (X3_DEFAULT_SHAPE == 16 ? 0xfff : 0)
#endif
#include "../pdgn_amd/csrc/gemm_x3.hip"

#include <cmath>
#include <cstdio>
#include <vector>

// the fp32 forms are not linked into this harness
int fp32_gemm_nt(long long, int, int, const float *, int, const float *, int, const float *, const float *, int, float *, int, float *, pdgn_stream_t) { return -1; }
int fp32_gemm_nn(long long, int, int, const float *, int, const float *, int, const float *, const float *, int, float *, int, float *, pdgn_stream_t) { return -1; }
int fp32_gemm_nt_ex(long long, int, int, const float *, int, const float *, int, const float *, const float *, int, float *, int, float *, const float *, int, int, int, const float *, int, int, pdgn_stream_t) { return -1; }
int fp32_gemm_tn_big(long long, int, int, const float *, int, const float *, int, float *, pdgn_stream_t) { return -1; }
long long fp32_gemm_nt_stat_rows(long long, int, int) { return -1; }
int fp32_gemm_nt_stat_block_rows(long long, int, int) { return -1; }
int fp32_gemm_nt_config(long long, int, int, int) { return -1; }

int main() {
    void *slots = nullptr;                                         // mode 2: the ring of operand-scale slots is the caller's
    hipMalloc(&slots, 4 << 20);
    hipMemset(slots, 0, 4 << 20);
    pdgn_gemm_set_scale_slots(slots, 4 << 20);
    const long long shapes[][3] = {{35840, 512, 5120}, {35840, 12832, 128}, {71680, 1024, 256}, {17920, 256, 2560}};
    printf("X3_ABLATE=%d mode %d CFG=%s |", X3_ABLATE, pdgn_gemm_set_mode(-1), getenv("PDGN_NT_CFG") ? getenv("PDGN_NT_CFG") : "-");
    for (auto &sh : shapes) {
        const long long m = sh[0];
        const int n = (int)sh[1], k = (int)sh[2];
        float *A, *W, *C;
        hipMalloc(&A, m * k * 4);
        hipMalloc(&W, (size_t)n * k * 4);
        hipMalloc(&C, m * n * 4);
        std::vector<float> h((size_t)n * k);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
        hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (long long r = 0; r < m; r += n) hipMemcpy(A + r * k, h.data(), (size_t)((m - r < n ? m - r : n)) * k * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) pdgn_gemm_nt(m, n, k, A, k, W, k, nullptr, nullptr, 0, C, n, nullptr, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) pdgn_gemm_nt(m, n, k, A, k, W, k, nullptr, nullptr, 0, C, n, nullptr, nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        // a few results against the host's fp64 (rows of A are rows of W: the harness's fill)
        double worst = 0.0;
        for (int t = 0; t < 48; ++t) {
            const long long i = (t * 7919LL) % m;
            const int j = (t * 104729) % n;
            float c;
            hipMemcpy(&c, C + i * n + j, 4, hipMemcpyDeviceToHost);
            double ref = 0.0, mag = 0.0;
            const float *ar = h.data() + (size_t)(i % n) * k, *wr = h.data() + (size_t)j * k;
            for (int q = 0; q < k; ++q) { ref += (double)ar[q] * wr[q]; mag += fabs((double)ar[q] * wr[q]); }
            const double e = fabs(c - ref) / (mag + 1e-30);
            worst = e > worst ? e : worst;
        }
        printf(" %lldx%dx%d %7.1f us %6.1f TF err %.2e |", m, n, k, ms * 100.f, 2.0 * m * n * k / (ms * 100.0) / 1e6, worst);
        hipFree(A); hipFree(W); hipFree(C);
    }
    printf("\n");
    return 0;
}
