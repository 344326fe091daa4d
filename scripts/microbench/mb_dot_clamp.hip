// mb_dot_clamp.hip -- does v_dot4_i32_i8 / v_dot2_i32_i16 with the clamp bit (the three-operand VOP3P form the compiler selects for
// __builtin_amdgcn_sdot4(..., clamp = true)) return the same integers as the two-operand v_dot4c form on gfx950? Random operands incl. negative bytes, and the
// "sc * dot" pattern of k_gemv_kq_ref.   build: hipcc --offload-arch=gfx950 -O3 -o mb_dot_clamp mb_dot_clamp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const int * a, const int * b, const int * c, int * o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = a[i], y = b[i], z = c[i];
    o[8 * i + 0] = __builtin_amdgcn_sdot4(x, y, 0, false);
    o[8 * i + 1] = __builtin_amdgcn_sdot4(x, y, 0, true);
    o[8 * i + 2] = __builtin_amdgcn_sdot4(x, y, z, false);
    o[8 * i + 3] = __builtin_amdgcn_sdot4(x, y, z, true);
    o[8 * i + 4] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x), __builtin_bit_cast(s2, y), z, false);
    o[8 * i + 5] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x), __builtin_bit_cast(s2, y), z, true);
    const int sc = (z & 63) - 32;
    o[8 * i + 6] = z + sc * __builtin_amdgcn_sdot4(x, y, 0, false);
    o[8 * i + 7] = z + sc * __builtin_amdgcn_sdot4(x, y, 0, true);
}
int main() {
    const int n = 1 << 20;
    int * h = (int *) malloc(3 * n * sizeof(int)), * ho = (int *) malloc(8 * n * sizeof(int));
    srand(7);
    for (int i = 0; i < 3 * n; ++i) h[i] = (rand() << 16) ^ rand() ^ (rand() << 31);
    for (int i = 2 * n; i < 3 * n; ++i) h[i] = (h[i] % (1 << 22));       // start values of the size the library uses
    int * d, * dout;
    hipMalloc(&d, 3 * n * sizeof(int)); hipMalloc(&dout, 8 * n * sizeof(int));
    hipMemcpy(d, h, 3 * n * sizeof(int), hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, d + n, d + 2 * n, dout, n);
    hipMemcpy(ho, dout, 8 * n * sizeof(int), hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0}, badref = 0;
    for (int i = 0; i < n; ++i) {
        int ref = 0; for (int j = 0; j < 4; ++j) ref += (int)(int8_t)(h[i] >> (8 * j)) * (int)(int8_t)(h[n + i] >> (8 * j));
        badref += ho[8 * i] != ref;
        for (int p = 0; p < 4; ++p) if (ho[8 * i + 2 * p] != ho[8 * i + 2 * p + 1]) { if (bad[p]++ < 3) printf("pair %d i %d: x %08x y %08x z %d: %d vs clamp %d\n", p, i, h[i], h[n + i], h[2 * n + i], ho[8 * i + 2 * p], ho[8 * i + 2 * p + 1]); }
    }
    printf("dot4c vs host: %ld bad; clamp form differs: dot4 zero-start %ld, dot4 start %ld, dot2 %ld, sc*dot %ld of %d\n", badref, bad[0], bad[1], bad[2], bad[3], n);
    return 0;
}
