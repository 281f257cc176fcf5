// How does v_mfma_f32_16x16x4_f32 add its four products to the accumulator?  One MFMA on random operands against
// candidate VALU evaluations of the same output, bit for bit:
//   seq  : fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))   (k = 0, 1, 2, 3, one rounding per step)
//   rev  : the same with k = 3, 2, 1, 0
//   pair : c + ((a0 b0 + a1 b1) + (a2 b2 + a3 b3)) with fma where possible
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_order.hip -o tools/ubench/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// lane = i + 16 kq: A[i][kq] and B[kq][j = i]; output regs r: D[4 (lane >> 4) + r][lane & 15]
__global__ __launch_bounds__(64) void k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                        float* __restrict__ d) {
    const int lane = threadIdx.x, blk = blockIdx.x;
    const float av = a[blk * 64 + lane], bv = b[blk * 64 + lane];
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[blk * 256 + lane * 4 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[blk * 256 + lane * 4 + r] = acc[r];
}
int main() {
    const int nblk = 4096;
    std::vector<float> A(nblk * 64), B(nblk * 64), C(nblk * 256), D(nblk * 256);
    srand(7);
    auto rnd = [](float s) { return s * ((float)rand() / RAND_MAX * 2.0f - 1.0f) * (1.0f + (rand() % 1000) * 1e-3f); };
    for (auto& v : A) v = rnd(1.0f);
    for (auto& v : B) v = rnd(100.0f);
    for (auto& v : C) v = rnd(10.0f);
    float *da, *db, *dc, *dd;
    if (hipMalloc(&da, A.size() * 4) || hipMalloc(&db, B.size() * 4) || hipMalloc(&dc, C.size() * 4) || hipMalloc(&dd, D.size() * 4)) return 3;
    if (hipMemcpy(da, A.data(), A.size() * 4, hipMemcpyHostToDevice) || hipMemcpy(db, B.data(), B.size() * 4, hipMemcpyHostToDevice) ||
        hipMemcpy(dc, C.data(), C.size() * 4, hipMemcpyHostToDevice)) return 3;
    hipLaunchKernelGGL(k, dim3(nblk), dim3(64), 0, 0, da, db, dc, dd);
    if (hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost)) return 3;
    long n = 0, m_seq = 0, m_rev = 0, m_pair = 0, m_nofma = 0;
    for (int blk = 0; blk < nblk; ++blk)
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * (lane >> 4) + r, j = lane & 15;       // output row (A's row), column (B's column)
                float av[4], bv[4];
                for (int kq = 0; kq < 4; ++kq) {
                    av[kq] = A[blk * 64 + i + 16 * kq];
                    bv[kq] = B[blk * 64 + j + 16 * kq];
                }
                const float c0 = C[blk * 256 + lane * 4 + r], got = D[blk * 256 + lane * 4 + r];
                float s = c0;
                for (int kq = 0; kq < 4; ++kq) s = __builtin_fmaf(av[kq], bv[kq], s);
                float t = c0;
                for (int kq = 3; kq >= 0; --kq) t = __builtin_fmaf(av[kq], bv[kq], t);
                const float p01 = __builtin_fmaf(av[1], bv[1], av[0] * bv[0]), p23 = __builtin_fmaf(av[3], bv[3], av[2] * bv[2]);
                const float u = c0 + (p01 + p23);
                float w = c0;
                for (int kq = 0; kq < 4; ++kq) w = w + av[kq] * bv[kq];
                ++n;
                m_seq += s == got;
                m_rev += t == got;
                m_pair += u == got;
                m_nofma += w == got;
            }
    printf("outputs %ld: sequential-fma k=0..3 matches %ld, k=3..0 %ld, pairwise %ld, mul-then-add k=0..3 %ld\n", n, m_seq, m_rev, m_pair, m_nofma);
    return 0;
}
