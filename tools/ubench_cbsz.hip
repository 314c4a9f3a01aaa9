// probe: does v_mfma_f64_4x4x4_4b_f64 honour CBSZ / ABID (broadcast of one block's A operand to all four blocks)?
// D_b[i][j] = sum_k A_b[i][k] B_b[k][j];  A lane = 16 k + 4 b + i, B lane = 16 k + 4 b + j, D lane = 16 i + 4 b + j.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_cbsz.hip -o tools/bin/ubench_cbsz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void k(const double *a, const double *b, double *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID>
int run(const double *da, const double *db, double *dout, const std::vector<double> &ha, const std::vector<double> &hb)
{
    k<CBSZ, ABID><<<1, 64>>>(da, db, dout);
    std::vector<double> ho(64);
    if (hipMemcpy(ho.data(), dout, 512, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    // expectation with broadcast: block b uses A of block (CBSZ ? (ABID within group) : b)
    int ok = 0, okplain = 0;
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 4; ++b) for (int j = 0; j < 4; ++j) {
        const int nb = 1 << CBSZ;                 // blocks per broadcast group
        const int src = (b / nb) * nb + (ABID % nb);
        double e = 0, ep = 0;
        for (int kk = 0; kk < 4; ++kk) { e += ha[16 * kk + 4 * src + i] * hb[16 * kk + 4 * b + j]; ep += ha[16 * kk + 4 * b + i] * hb[16 * kk + 4 * b + j]; }
        ok += ho[16 * i + 4 * b + j] == e;
        okplain += ho[16 * i + 4 * b + j] == ep;
    }
    printf("cbsz %d abid %d: matches broadcast model %d/64, matches no-broadcast %d/64\n", CBSZ, ABID, ok, okplain);
    return 0;
}
int main()
{
    std::vector<double> ha(64), hb(64);
    for (int i = 0; i < 64; ++i) { ha[i] = 1 + i * 3 % 17; hb[i] = 2 + i * 5 % 13; }
    double *da, *db, *dout;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dout, 512);
    hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
    run<0, 0>(da, db, dout, ha, hb);
    run<2, 0>(da, db, dout, ha, hb); run<2, 1>(da, db, dout, ha, hb); run<2, 2>(da, db, dout, ha, hb); run<2, 3>(da, db, dout, ha, hb);
    run<1, 0>(da, db, dout, ha, hb); run<1, 1>(da, db, dout, ha, hb);
    return 0;
}
