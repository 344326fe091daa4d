// mb_mfma_f32_16.hip -- which f32 arithmetic does v_mfma_f32_16x16x4_f32 perform per output element? Candidate: one fused multiply-add
// per k step, k ascending (the arithmetic measured for v_mfma_f32_32x32x2_f32, mb_mfma_f32.hip):
//   seq_fma   c = fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, c))))
// Random inputs (including cancellation-heavy ones) through a chain of 16 MFMAs (K = 64), compared bit for bit with host-side chains.
// hipcc --offload-arch=gfx950 -O2 -o mb_mfma_f32_16 mb_mfma_f32_16.hip && ./mb_mfma_f32_16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
// A: [16 rows i][K], B: [K][16 cols j] -> D[16][16]. Assumed operand layout of 16x16x4: lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; D register r of lane l = D[i = r + 4 (l >> 4)][j = l & 15].
__global__ void k(const float * A, const float * B, float * D, int K) {
    const int l = threadIdx.x;
    v4f c = {0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[(l & 15) * K + k0 + (l >> 4)], b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(r + 4 * (l >> 4)) * 16 + (l & 15)] = c[r];
}
int main() {
    const int K = 64;
    std::vector<float> A(16 * K), B(K * 16), D(16 * 16);
    srand(7);
    auto rnd = [] { return (float)((rand() % 20001) - 10000) / 1000.0f * ((rand() & 7) == 0 ? 1e-3f : 1.0f) * ((rand() & 15) == 0 ? 1e3f : 1.0f); };
    int bad_seq = 0, bad_rev = 0, bad_pair = 0, bad_mul = 0, total = 0;
    float * dA, * dB, * dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    for (int rep = 0; rep < 8; ++rep) {
        for (auto & v : A) v = rnd();
        for (auto & v : B) v = rnd();
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float seq = 0, rev = 0, pair = 0, mul = 0;
            for (int k0 = 0; k0 < K; k0 += 4) {
                float p[4];
                for (int q = 0; q < 4; ++q) p[q] = 0;
                for (int q = 0; q < 4; ++q) seq = fmaf(A[i * K + k0 + q], B[(k0 + q) * 16 + j], seq);
                for (int q = 3; q >= 0; --q) rev = fmaf(A[i * K + k0 + q], B[(k0 + q) * 16 + j], rev);
                float t = 0; for (int q = 0; q < 4; ++q) t = fmaf(A[i * K + k0 + q], B[(k0 + q) * 16 + j], t);
                pair = pair + t;
                for (int q = 0; q < 4; ++q) mul = mul + A[i * K + k0 + q] * B[(k0 + q) * 16 + j];
                (void) p;
            }
            const float d = D[i * 16 + j];
            bad_seq += memcmp(&d, &seq, 4) != 0; bad_rev += memcmp(&d, &rev, 4) != 0; bad_pair += memcmp(&d, &pair, 4) != 0; bad_mul += memcmp(&d, &mul, 4) != 0;
            ++total;
        }
    }
    printf("v_mfma_f32_16x16x4_f32 over K = %d, %d outputs: mismatches vs  seq_fma %d   rev_fma %d   group_then_add %d   mul_add(unfused) %d\n", K, total, bad_seq, bad_rev, bad_pair, bad_mul);
    return bad_seq != 0;
}
