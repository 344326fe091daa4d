// mb_mfma_f32.hip -- which f32 arithmetic does v_mfma_f32_32x32x2_f32 perform per output element? Candidates, for the two
// k-steps of one instruction (a0 b0, a1 b1) added to c:
//   seq_fma   fmaf(a1, b1, fmaf(a0, b0, c))            one fused multiply-add per k step, k ascending
//   rev_fma   fmaf(a0, b0, fmaf(a1, b1, c))
//   pair_add  c + (a0 b0 + a1 b1) in various roundings
// The program runs random inputs (including cancellation-heavy ones) through a chain of 32 MFMAs (K = 64) and compares the
// result bit for bit with host-side chains. hipcc --offload-arch=gfx950 -O2 -o mb_mfma_f32 mb_mfma_f32.hip && ./mb_mfma_f32
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
// A: [32 rows i][K], B: [K][32 cols j] -> D[32][32]. Operand layout of 32x32x2: lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; D register r of lane l = D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31].
__global__ void k(const float * A, const float * B, float * D, int K) {
    const int l = threadIdx.x;
    v16f c = {0};
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[(l & 31) * K + k0 + (l >> 5)], b = B[(k0 + (l >> 5)) * 32 + (l & 31)];
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
    const int K = 64;
    std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
    srand(7);
    auto rnd = [] { return (float)((rand() % 20001) - 10000) / 1000.0f * ((rand() & 7) == 0 ? 1e-3f : 1.0f) * ((rand() & 15) == 0 ? 1e3f : 1.0f); };
    for (auto & v : A) v = rnd();
    for (auto & v : B) v = rnd();
    float * dA, * dB, * dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_rev = 0, bad_pair = 0, bad_mul = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float seq = 0, rev = 0, pair = 0, mul = 0;
        for (int k0 = 0; k0 < K; k0 += 2) {
            const float a0 = A[i * K + k0], a1 = A[i * K + k0 + 1], b0 = B[k0 * 32 + j], b1 = B[(k0 + 1) * 32 + j];
            seq = fmaf(a1, b1, fmaf(a0, b0, seq));
            rev = fmaf(a0, b0, fmaf(a1, b1, rev));
            pair = pair + fmaf(a1, b1, a0 * b0);
            mul = (mul + a0 * b0) + a1 * b1;
        }
        const float d = D[i * 32 + j];
        bad_seq += memcmp(&d, &seq, 4) != 0; bad_rev += memcmp(&d, &rev, 4) != 0; bad_pair += memcmp(&d, &pair, 4) != 0; bad_mul += memcmp(&d, &mul, 4) != 0;
    }
    printf("v_mfma_f32_32x32x2_f32 over K = %d, 1024 outputs: mismatches vs  seq_fma %d   rev_fma %d   pair_add %d   mul_add(unfused) %d\n", K, bad_seq, bad_rev, bad_pair, bad_mul);
    return 0;
}
