// hx_sht.hip -- HEALPix spherical harmonic transforms for gfx950.
//
// Replaces healpy.map2alm / alm2map as called from heracles/healpy.py:183-189 (spin 0 and
// spin 2, RING-ordered maps, mmax == lmax).
//
// Pipeline of one analysis sweep (<= 16 map components):
//   1. k_ring_subdft       ring Fourier stage.  A north/south ring pair is packed as
//                          z = f_N + i f_S and transformed as one complex DFT of length
//                          4n, split radix-4 (DIF) into four length-n DFTs; a work item is ONE of
//                          them, run entirely in LDS (fused radix-8 / radix-16 passes on a padded
//                          buffer; plain FFT for n = 2^k, Bluestein otherwise), the four items of a
//                          ring pair in four work-groups of one XCD so that they share its L2.
//   2. k_fourier_combine   un-packs N/S, applies ring phase / quadrature weight, forms
//                          the parity combinations and writes the MFMA B-operand layout
//                          F[m][ring pair][parity][op][columns].
//   3. k_legendre_pipe     (hx_analysis.hip) Legendre / Wigner-d stage: lanes = ring pairs run the
//                          three-term recursion in l; tiles of lambda_lm go through LDS into A operands
//                          of v_mfma_f64_16x16x4_f64, which contracts over rings against the F operands
//                          held in registers; one work-group per m adds its ring groups in place.
//   4. k_alm_reduce        rows -> alm layout (x fl); for the small-batch kernels also the fixed-order
//                          sum of the ring-group partials.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include <type_traits>

#include "hx_sht_common.h"

// (measured and not kept, round 5: the ring spectra leaving with non-temporal stores -- profiles/r05_fft_cycles.txt)

namespace hx {
using namespace hxfft;

// =====================================================================================
// table initialisation kernels
// =====================================================================================
// Normalised recursions used by the analysis kernel (two FMAs per new value, after the
// scheme of libsharp/ducc's Ylmgen): with lambda_l = alpha_l mu_l,
//   spin 0 (two-step):  mu_{l+2} = (A' x^2 + B') mu_l - mu_{l-2}
//       A = a_{l+1} a_{l+2},  B = -(a_{l+2}/a_{l+1} + a_{l+2} a_{l+1}/a_l^2),  a_l = sqrt((4l^2-1)/(l^2-m^2)),
//       alpha_{l+2} alpha_l = a_{l+2} a_{l+1} / 4,   A' = A alpha_l/alpha_{l+2},  B' likewise;
//       coef[idx(l,m)] = (A', B') is indexed by the SOURCE l, alpha[idx(l,m)] = alpha_l.
//   spin 2 (one-step):  mu_{l+1} = (p' x +- q') mu_l - mu_{l-1}   (+ for d^l_{m,-2}, - for d^l_{m,+2})
//       alpha_{l+1} = r_l alpha_{l-1},  p' = p alpha_l/alpha_{l+1},  q' likewise;
//       coef[idx(l+1,m)] = (p', q') is indexed by the TARGET l.
// One thread per (m, chain): the alpha recursion is sequential in l.
__global__ void k_init_norm0(int lmax, double2 *__restrict__ coef, double *__restrict__ alpha)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = t >> 1, par = t & 1;
    if (m > lmax) return;
    const double dm = m;
    auto a = [dm](double l) { return sqrt((4.0 * l * l - 1.0) / (l * l - dm * dm)); };
    double al = 1.0;
    for (int l = m + par; l <= lmax; l += 2) {
        const double dl = l;
        const double a1 = a(dl + 1.0), a2 = a(dl + 2.0);
        const double A = a1 * a2;
        double B = -a2 / a1;
        if (l > m) {
            const double a0 = a(dl);
            B -= a2 * a1 / (a0 * a0);
        }
        const double an = 0.25 * a2 * a1 / al;
        coef[almidx(lmax, l, m)] = make_double2(A * al / an, B * al / an);
        alpha[almidx(lmax, l, m)] = al;
        al = an;
    }
}

__global__ void k_init_norm2(int lmax, double2 *__restrict__ coef, double *__restrict__ alpha)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m > lmax) return;
    const int l0 = m > 2 ? m : 2;
    double am1 = 1.0, a0 = 1.0;  // alpha_{l-1}, alpha_l
    for (int l = l0; l <= lmax; ++l) {
        alpha[almidx(lmax, l, m)] = a0;
        if (l == lmax) break;
        // coefficients of the step l -> l+1 (n = -2)
        const double k = l, lp = l + 1.0, dm = m, dn = -2.0;
        const double den = k * sqrt((lp * lp - dm * dm) * (lp * lp - dn * dn));
        const double r1 = sqrt((2.0 * k + 3.0) / (2.0 * k + 1.0));
        const double p = r1 * (2.0 * k + 1.0) * k * lp / den;
        const double q = -r1 * (2.0 * k + 1.0) * dm * dn / den;
        double a1 = 1.0;
        if (l > l0) {
            const double r2 = sqrt((2.0 * k + 3.0) / (2.0 * k - 1.0));
            const double r = r2 * lp * sqrt((k * k - dm * dm) * (k * k - dn * dn)) / den;
            a1 = r * am1;
        }
        coef[almidx(lmax, l + 1, m)] = make_double2(p * a0 / a1, q * a0 / a1);
        am1 = a0;
        a0 = a1;
    }
}

// Bluestein filter spectra, one block per ring pair whose sub-length is not a power of two
// and is the first ring with that length.  The spectrum H (bit-reversed order, as the forward passes leave it) is stored
// TRANSPOSED, bhat[j (M/16) + i] = H[16 i + j]: in the ring kernel the thread that owns elements 16 i .. 16 i + 15 after the
// last forward pass multiplies them in registers, and for a fixed j consecutive threads then read consecutive entries
// (M < 16 -- the ring of 12 pixels -- keeps the plain order).
__global__ __launch_bounds__(512) void k_init_bhat(PlanDev P, const int *__restrict__ rp_list,
                                                   double2 *__restrict__ bhat)
{
    extern __shared__ double2 buf[];
    const int rp = rp_list[blockIdx.x];
    const int n = P.nsub[rp];
    const int M = fft_size_for(n);
    for (int j = threadIdx.x; j < lds_fft_slots(M); j += blockDim.x) buf[j] = make_double2(0.0, 0.0);
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double2 c = expipi((double)chirp_num(j, n) / (double)n);
        buf[lds_slot(j)] = c;
        if (j) buf[lds_slot(M - j)] = c;
    }
    __syncthreads();
    lds_fft_dif(buf, M, P.tw, P.twN);
    double2 *out = bhat + P.bhat_off[rp];
    for (int e = threadIdx.x; e < M; e += blockDim.x) out[M >= 16 ? (e & 15) * (M >> 4) + (e >> 4) : e] = buf[lds_slot(e)];
}

// =====================================================================================
// 1. ring Fourier stage: sub-DFTs in LDS
// =====================================================================================
// MODE 0: input = real maps (N ring -> real part, S ring -> imaginary part)
// MODE 1: input = complex spectrum Zc[c][ny-layout natural order] (synthesis: conj trick)
//
// A work item is ONE of the four length-n sub-DFTs X[4k + r] of a (ring pair, component): the group reads the 4n packed
// pixels z_q[j] = z[j + q n], forms t_r[j] = sum_q z_q[j] (-i)^(q r) on the fly and runs one transform in a padded LDS buffer,
// M / 16 threads, one radix-16 butterfly each per pass.  The four items of a ring pair are dealt to four work-groups of the
// SAME XCD that run at the same time (items are numbered xcd-minor, groups are persistent and walk items b, b + G, ...), so
// the pixels come from HBM once and from that XCD's L2 three more times -- and no thread has to keep a ring's pixels in
// registers across four transforms, which is what held the first version at one wave per SIMD with every latency exposed
// (256 registers of pixels beside the butterfly).  A transform is 3-4 LDS round trips (fused radix-8 / radix-16 passes); a
// Bluestein convolution fuses the last forward pass, the filter and the first inverse pass in registers (they work on the
// same 16 consecutive elements): 5-7 round trips for what were 13-15 with radix-4 passes and a filter pass of its own.  The
// filter values of a thread's butterfly are requested before the forward passes.  Every phase factor exp(-i pi q / 2n) (load
// phase, Bluestein chirps) is hi[q >> 6] lo[q & 63] from two small tables built per item in LDS: (4n / 64 + 65) sincos per
// item instead of one per pixel.
__host__ __device__ inline int ring_ph_hi(int M) { return (4 * M) / 64 + 1; }  // entries of the coarse phase table: q >> 6 for q < 4n, n <= M
constexpr int RING_NTMAX = 512;  // threads per group: M / 16 (one radix-16 butterfly per thread and pass), 64 at least
constexpr int RING_FB = 8;       // values of j per thread whose pixel loads are in flight together (64 loads)

// WSYM (MODE 0): the pixel-weight array has the symmetry of healpy's full weights -- it repeats over the four quadrants of a ring
// and from the northern to the southern ring of a pair -- so ONE weight per pixel pair of the first quadrant is read instead of
// eight.  Whether an array has that symmetry is found once per call of the C ABI (k_pixw_symmetry, one read of the array and a
// 4-byte read-back before anything else of the call is queued).  A template parameter, because a second run-time branch inside
// the batches of loads splits them (17.2 instead of 14.0 ms per 8 components even without weights); "weights or none" stays the
// run-time test it was (as a compile-time constant the 64 loads of the generic path are issued together: 256 registers, 34 spilled).
template <int MODE, bool WSYM = false>
__global__ __launch_bounds__(RING_NTMAX) void k_ring_subdft(PlanDev P, const RingDesc *__restrict__ desc, int nrings, int nb, int Mclass,
                                                            const double *__restrict__ maps,
                                                            const double *__restrict__ pixw,
                                                            const double2 *__restrict__ zin,
                                                            double2 *__restrict__ Y, double *__restrict__ pixout = nullptr,
                                                            const double *__restrict__ ref = nullptr)
{
    extern __shared__ double2 buf[];  // the padded transform buffer of the class's M, then the phase tables (4 M / 64 + 1 and 64 entries)
    __shared__ double2 tw_hi[TW_HI_MAX], tw_lo[64];
    double2 *ph_hi = buf + lds_fft_slots(Mclass), *ph_lo = ph_hi + ring_ph_hi(Mclass);
    const int nt = blockDim.x;
    const int nitems = ((nrings + 7) >> 3) * nb * 32;  // sets of 8 ring pairs (one per XCD) x components x 4 sub-DFTs
    const TwFactored twf = load_tw_factored(tw_hi, tw_lo, P.tw, P.twN);  // visible after the first barrier below
    int tid = threadIdx.x;
    // What an item needs to know about its ring pair is ONE 32-byte record (RingDesc; it was rp_list[ring], then P.nsub / startN / startS /
    // bhat_off [rp]: two dependent trips to memory in front of the pixel loads, a third one -- cycle accounting, profiles/r04_fft_cycles.txt:
    // "load + tables" 24k cycles per item whatever the length of its ring), and the record of the NEXT item is requested behind the first
    // batch of pixel loads of this one -- by a vector load with the same address in every lane (a scalar load would make the first
    // lgkmcnt(0) of the item wait for it) -- and moved to scalars after the fill, when it has long landed.
    auto ring_of = [&](int item) __attribute__((always_inline)) { return ((item >> 5) / nb) * 8 + (item & 7); };
    RingDesc cur = RingDesc{0, 0, 0, 1, 0};
    if ((int)blockIdx.x < nitems && ring_of(blockIdx.x) < nrings) cur = desc[ring_of(blockIdx.x)];
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        // item = 8 (4 s + r) + x: sub-DFT r of set s on XCD x -- the four r of a (ring pair, component) in four groups of that XCD,
        // side by side in time.  Set s = (ring set s / nb, component s % nb): an XCD walks the COMPONENTS of one ring pair before it
        // moves to its next ring pair, so that the pair's pixel weights come from HBM once and from that XCD's L2 for every other
        // component (component-major sets re-read the 1.6 GB weight array per component: +34 ms per step of the bench)
        const int r = (item >> 3) & 3, set = item >> 5;
        const int ring = ring_of(item), c = set % nb;
        const int itn = item + gridDim.x, ringn = itn < nitems ? ring_of(itn) : nrings;
        const bool nextv = ringn < nrings;
        if (ring >= nrings) {  // padding of the last set of 8 ring pairs
            if (nextv) cur = desc[ringn];
            continue;
        }
        const int n = cur.n;
        const long long sN = cur.sN, sS = cur.sS;
        const int M = fft_size_for(n), MP = lds_fft_slots(M);
        const bool blu = M != n;
        const int p = ilog2(M);
        const double2 *bh = P.bhat + cur.bhat_off;
        const double inv = 1.0 / M, inv4n = 0.25 / (double)n;
        // the thread index goes through an empty asm statement per item, so that what derives from it (LDS addresses, pixel
        // offsets) is set up per item instead of being hoisted out of this loop and kept in registers across the passes
        asm volatile("; item" : "+v"(tid));
        // ---- pixels.  RING_FB values of j at a time: all their loads (8 per j) are issued before the first is used -- one trip
        // to L2 / HBM per batch instead of one per j (out-of-range j read pixel 0 of the ring and write the spare slot of the
        // buffer; a missing southern ring (the equator) reads the northern one and selects 0).  The first batch is requested
        // before the phase tables are built, whose sincospi then cover part of its way ----
        const bool haveS = sS >= 0, odd = r & 1, pw = MODE == 0 && !WSYM && pixw != nullptr;
        constexpr bool wsym = MODE == 0 && WSYM;
        const double sg = (r & 2) ? -1.0 : 1.0;
        const double *mpN = maps + (long long)c * P.npix + sN, *mpS = maps + (long long)c * P.npix + (haveS ? sS : sN);
        const double *pwN = pixw + sN, *pwS = pixw + (haveS ? sS : sN);
        const double2 *zp = zin + (long long)c * P.ny + sN;
        double2 z[RING_FB][4];
        // A thread takes PAIRS of neighbouring j (j = 2 p, 2 p + 1, p = tid + k nt): one 16-byte load per ring, segment q and pair
        // instead of two 8-byte ones -- with pixel weights a batch is 64 loads, not 128 (the fill is bound by the number of load
        // instructions in flight, not by bytes).  An odd n leaves a last pair of one element: it reads the pair before it and
        // shifts (the segment [q n, (q + 1) n) is followed by the next one -- or, for q = 3, by the next ring, which the last
        // ring of the map does not have).
        struct __attribute__((aligned(8))) Pair { double x, y; };
        // A batch is REQUESTED here and FINISHED (pixel weights of the symmetric kind, the odd-n tail, the missing southern ring) where the
        // fill uses it: z[u][q], z[u + 1][q] hold the raw northern and southern pair until then.  Finished at the load -- as it was
        // since the weights came into the path -- every product needs its operand at once and hipcc issued two loads, s_waitcnt vmcnt(0),
        // two loads, ...: sixteen trips to memory one after the other, 23k of an item's 57-75k cycles whatever the length of its ring
        // (profiles/r04_fft_cycles.txt).  (Generic weight arrays keep that form: their raw values would need another 128 registers.)
        Pair wsy[RING_FB / 2];
        auto load_batch_t = [&](auto WIDEC, int u0) __attribute__((always_inline)) {
            constexpr bool WIDE = decltype(WIDEC)::value;
#pragma unroll
            for (int u = 0; u < RING_FB; u += 2) {
                const int j = 2 * (tid + ((u0 + u) >> 1) * nt);      // first j of the pair; the batch covers j < (u0 + RING_FB) nt
                const bool tail = WIDE && j == n - 1;                // (n = 1: the scalar path below)
                const int jj = tail ? n - 2 : (j + 1 < n ? j : 0);
                wsy[u >> 1] = Pair{1.0, 1.0};
                if (MODE == 0 && wsym && WIDE) wsy[u >> 1] = *reinterpret_cast<const Pair *>(pwN + jj);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = jj + q * n;
                    if (MODE == 0) {
                        Pair fn, fs;
                        if (WIDE) {
                            fn = *reinterpret_cast<const Pair *>(mpN + i);
                            fs = *reinterpret_cast<const Pair *>(mpS + i);
                            if (!wsym && pw) {
                                const Pair wn = *reinterpret_cast<const Pair *>(pwN + i), ws = *reinterpret_cast<const Pair *>(pwS + i);
                                fn.x *= wn.x; fn.y *= wn.y; fs.x *= ws.x; fs.y *= ws.y;
                            }
                            z[u][q] = make_double2(fn.x, fn.y);      // raw: finish_batch
                            z[u + 1][q] = make_double2(fs.x, fs.y);
                        } else {
                            fn.x = fn.y = mpN[q]; fs.x = fs.y = mpS[q];
                            if (pw || wsym) { fn.x *= pwN[q]; fs.x *= pwS[q]; fn.y = fn.x; fs.y = fs.x; }
                            z[u][q] = make_double2(fn.x, haveS ? fs.x : 0.0);
                            z[u + 1][q] = make_double2(fn.y, haveS ? fs.y : 0.0);
                        }
                    } else {
                        const int i0 = (j < n ? j : 0) + q * n, i1 = (j + 1 < n ? j + 1 : 0) + q * n;
                        z[u][q] = zp[i0];
                        z[u + 1][q] = zp[i1];
                    }
                }
            }
        };
        auto load_batch = [&](int u0) __attribute__((always_inline)) {
            if (n >= 2) load_batch_t(std::true_type{}, u0);
            else load_batch_t(std::false_type{}, u0);
        };
        // raw pairs -> the values of j and j + 1: (north, south) each
        auto finish_batch = [&](int u0) __attribute__((always_inline)) {
            if (MODE != 0 || n < 2) return;
#pragma unroll
            for (int u = 0; u < RING_FB; u += 2) {
                const int j = 2 * (tid + ((u0 + u) >> 1) * nt);
                const bool tail = j == n - 1;
                const Pair w = wsy[u >> 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double2 rn = z[u][q], rs = z[u + 1][q];
                    const double nx = wsym ? rn.x * w.x : rn.x, ny = wsym ? rn.y * w.y : rn.y;
                    const double sx = wsym ? rs.x * w.x : rs.x, sy = wsym ? rs.y * w.y : rs.y;
                    z[u][q] = make_double2(tail ? ny : nx, haveS ? (tail ? sy : sx) : 0.0);
                    z[u + 1][q] = make_double2(ny, haveS ? sy : 0.0);
                }
            }
        };
        load_batch(0);
        int4 nd0 = make_int4(0, 0, 0, 0), nd1 = nd0;  // the next item's record (see above)
        if (nextv) {
            const int4 *dp = reinterpret_cast<const int4 *>(desc + ringn) + (tid >> 30);
            nd0 = dp[0];
            nd1 = dp[1];
        }
        // (work-group barriers that wait for this wave's LDS traffic only: __syncthreads() would wait for the pixel loads too)
        auto lds_barrier = []() __attribute__((always_inline)) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        };
        lds_barrier();  // the previous item's last readers of the phase tables and of the buffer
        for (int a = tid; a <= (4 * n) >> 6; a += nt) ph_hi[a] = expipi(-(double)(a << 6) / (2.0 * n));
        if (tid < 64) ph_lo[tid] = expipi(-(double)tid / (2.0 * n));
        lds_barrier();
        auto phase = [&](unsigned q) __attribute__((always_inline)) {  // exp(-i pi q / 2n), q < 4n
            return cmul(ph_hi[q >> 6], ph_lo[q & 63]);
        };
        // ---- fill: t_r[j] x load phase.  t_r = (a, c)[r & 1] +- (b, d)[r & 1] with a = z0 + z2, b = z1 + z3, c = z0 - z2,
        // d = -i (z1 - z3) ----
        {
            for (int u0 = 0;;) {
                finish_batch(u0);
#pragma unroll
                for (int u = 0; u < RING_FB; ++u) {
                    const int j = 2 * (tid + ((u0 + u) >> 1) * nt) + (u & 1);  // the pairs of load_batch
                    const double2 e0 = odd ? csub(z[u][0], z[u][2]) : cadd(z[u][0], z[u][2]);
                    const double2 e1 = odd ? mul_mi(csub(z[u][1], z[u][3])) : cadd(z[u][1], z[u][3]);
                    const double2 t = make_double2(fma(sg, e1.x, e0.x), fma(sg, e1.y, e0.y));
                    // (j r + 2 j^2 [Bluestein]) mod 4n = load_phase_num(j, r, n, blu); j < 2^14: 32 bits hold it
                    const unsigned jc = j < n ? j : 0;
                    const unsigned qn = blu ? mod_by_inv(jc * (unsigned)r + 2u * jc * jc, 4u * (unsigned)n, inv4n) : jc * (unsigned)r;  // (j r < 4n as it is)
                    buf[j < n ? lds_slot(j) : MP - 1] = cmul(t, phase(qn));
                }
                u0 += RING_FB;
                if (u0 * nt >= n) break;
                load_batch(u0);
            }
            if (blu)
                for (int j = n + tid; j < M; j += nt) buf[lds_slot(j)] = make_double2(0.0, 0.0);  // Bluestein padding
        }
        __syncthreads();
        if (nextv) {
            auto sc = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
            cur.sN = ((long long)sc(nd0.y) << 32) | (unsigned)sc(nd0.x);
            cur.sS = ((long long)sc(nd0.w) << 32) | (unsigned)sc(nd0.z);
            cur.bhat_off = ((long long)sc(nd1.y) << 32) | (unsigned)sc(nd1.x);
            cur.n = sc(nd1.z);
            cur.rp = sc(nd1.w);
        }
        double2 *out = Y + (long long)c * P.ny + sN + (long long)r * n;
        // MODE 1 with `pixout` (synthesis, round 6): the value of bin k of sub-DFT r IS the pixel pair 4 k + r of the two rings --
        // Y_r[k] = conj(z[4 k + r]), f_N = Re, f_S = -Im -- so it goes straight to the maps (or, with `ref`, the residual ref - synthesised of
        // a Jacobi iteration) instead of through Y and a scatter pass of its own (16 B written + 16 B read + 16 B written per pixel pair
        // before; the four items of a ring pair run side by side on one XCD, whose L2 merges their interleaved 8-byte stores)
        double *pxN = nullptr, *pxS = nullptr;
        const double *rfN = nullptr, *rfS = nullptr;
        if (MODE == 1 && pixout) {
            pxN = pixout + (long long)c * P.npix + sN + r;
            pxS = pixout + (long long)c * P.npix + (haveS ? sS : sN) + r;
            if (ref) { rfN = ref + (long long)c * P.npix + sN + r; rfS = ref + (long long)c * P.npix + (haveS ? sS : sN) + r; }
        }
        // MODE 0: bins 4 k + r that no order m <= lmax falls on (neither as m nor as 4n - m) are not written: a ring of 4n > 2 lmax + 1
        // pixels leaves lmax < bin < 4n - lmax out -- a quarter of the belt's stores at nside 4096 / lmax 6144, half at nside 8192 / lmax 8000
        // (the read-out is a burst of stores into the in-order memory pipeline: what the item waits for at its end)
        const int kdrop0 = MODE == 0 ? (P.lmax - r) / 4 + 1 : n, kdrop1 = (4 * n - P.lmax - r + 3) / 4;
        auto emit = [&](int k, double2 v) __attribute__((always_inline)) {
            if (MODE == 0 && k >= kdrop0 && k < kdrop1) return;
            if (MODE == 1 && pxN) {
                double fn = v.x, fs = -v.y;
                if (rfN) { fn = rfN[4 * k] - fn; fs = rfS[4 * k] - fs; }
                pxN[4 * k] = fn;
                if (haveS) pxS[4 * k] = fs;
            } else {
                out[k] = v;
            }
        };
        if (!blu) {
            lds_fft_dif(buf, M, twf, P.twN);
            for (int k = tid; k < n; k += nt) emit(k, buf[lds_slot(bitrev(k, p))]);
            continue;
        }
        if (M >= 16) {
            // filter values of this thread's first butterfly of the fused pass: requested before the forward passes, which
            // cover the trip to L2 / HBM (waited for inside the butterfly loop it cost 13 000 cycles per butterfly)
            double2 bq[16];
            const int nbf = M >> 4;
#pragma unroll
            for (int j = 0; j < 16; ++j) bq[j] = bh[j * nbf + (tid < nbf ? tid : 0)];
            lds_fft_dif(buf, M, twf, P.twN, tid, nt, true);
            // last forward pass (h = 1: no twiddles), filter, first inverse pass on the thread's 16 consecutive elements
#pragma unroll 1
            for (int i = tid; i < nbf; i += nt) {
                double2 x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = buf[lds_slot(16 * i) + j];
                dif_regs<4>(x);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x[j] = cmul(x[j], bq[j]);
                }
                if (i + nt < nbf) {  // (groups of fewer than M / 16 threads: tiny rings only)
#pragma unroll
                    for (int j = 0; j < 16; ++j) bq[j] = bh[j * nbf + i + nt];
                }
                dit_inv_regs<4>(x);
#pragma unroll
                for (int j = 0; j < 16; ++j) buf[lds_slot(16 * i) + j] = x[j];
            }
            __syncthreads();
            lds_fft_dit_inv(buf, M, twf, P.twN, tid, nt, true);
        } else {
            lds_fft_dif(buf, M, twf, P.twN, tid, nt);
            for (int j = tid; j < M; j += nt) buf[lds_slot(j)] = cmul(buf[lds_slot(j)], bh[j]);
            __syncthreads();
            lds_fft_dit_inv(buf, M, twf, P.twN, tid, nt);
        }
        for (int k = tid; k < n; k += nt)  // chirp exp(-i pi k^2 / n)
            emit(k, cscale(cmul(buf[lds_slot(k)], phase(2u * mod_by_inv((unsigned)k * (unsigned)k, 2u * (unsigned)n, 2.0 * inv4n))), inv));
    }
}

// =====================================================================================
// 1a. rings of 4 x 2^k pixels (the equatorial belt, and the cap rings with n = 2^k): plain FFTs, no Bluestein convolution.  The
//     four work items of a (ring pair, component) of the kernel above each read ALL of the pair's pixels (and pixel weights): 3 of
//     the 4 reads come from L2, but the belt is 5120 of the 8192 ring pairs at nside 4096 and its items spend 60 % (without weights)
//     to 80 % (with) of their cycles waiting for those reads (tools/fft_ablate.sh 32: load + fill 102k of 173k / 275k of 337k
//     cycles per ring pair and component) -- 515 GB through the L2s per step of the bench.  Here a work item is TWO sub-DFTs,
//     r and r + 2: they are the sum and the difference of the same two combinations of the four segments
//         t_r = e0 + e1,  t_{r+2} = e0 - e1,   e0 = z0 +- z2,  e1 = z1 +- z3 (x -i for odd r),
//     so one read of the pixels fills two LDS buffers, and the two halves of the work-group (M / 16 threads each) run one
//     transform each: half the reads.  Two 4096-point buffers are 139 KiB: one group of 512 threads per CU.
// =====================================================================================
template <int MODE, bool WSYM = false>
__global__ __launch_bounds__(RING_NTMAX) void k_ring_pairfft(PlanDev P, const RingDesc *__restrict__ desc, int nrings, int nb, int M,
                                                             const double *__restrict__ maps, const double *__restrict__ pixw,
                                                             const double2 *__restrict__ zin, double2 *__restrict__ Y,
                                                             double *__restrict__ pixout = nullptr, const double *__restrict__ ref = nullptr)
{
    extern __shared__ double2 buf[];  // two padded buffers of M points, then the phase tables (4 M / 64 + 1 and 64 entries)
    __shared__ double2 tw_hi[TW_HI_MAX], tw_lo[64];
    const int MP = lds_fft_slots(M), n = M, p = ilog2(M);
    double2 *ph_hi = buf + 2 * MP, *ph_lo = ph_hi + ring_ph_hi(M);
    const int nt = blockDim.x, nh = nt >> 1;  // threads of the group / of one transform
    // Two work items per (ring pair, component), one per round, side by side on one XCD.  (Measured and not kept: one work item that runs
    // both rounds from ONE read, its pixels -- 128 registers -- kept across the transforms: 46 registers spilled, 18.9 vs 19.4 ms per 8
    // components with pixel weights, 15.1 vs 14.3 without, same device.)
    const int nitems = ((nrings + 7) >> 3) * nb * 16;
    const TwFactored twf = load_tw_factored(tw_hi, tw_lo, P.tw, P.twN);
    int tid = threadIdx.x;
    // the phase tables depend on n = M only: built once per group
    for (int a = tid; a <= (4 * n) >> 6; a += nt) ph_hi[a] = expipi(-(double)(a << 6) / (2.0 * n));
    if (tid < 64) ph_lo[tid] = expipi(-(double)tid / (2.0 * n));
    __syncthreads();
    auto phase = [&](unsigned q) __attribute__((always_inline)) { return cmul(ph_hi[q >> 6], ph_lo[q & 63]); };  // exp(-i pi q / 2n), q < 4n
    // (the record of the next item's ring pair is requested behind this item's pixel loads: see k_ring_subdft)
    auto ring_of = [&](int item) __attribute__((always_inline)) { return ((item >> 4) / nb) * 8 + (item & 7); };
    auto comp_of = [&](int item) __attribute__((always_inline)) { return (item >> 4) % nb; };
    RingDesc cur = RingDesc{0, 0, 0, 1, 0};
    if ((int)blockIdx.x < nitems && ring_of(blockIdx.x) < nrings) cur = desc[ring_of(blockIdx.x)];
    // Pixels: pairs of neighbouring j: 2 (tid + k nt), k < 4 (n / 2 pairs over nt = n / 8 threads, or 128 threads for n <= 1024).  A batch is
    // REQUESTED raw -- z[u][q], z[u + 1][q] hold the northern and the southern pair, the symmetric weight pair sits beside them -- and
    // FINISHED where the fill uses it (k_ring_subdft).  The FIRST HALF of the NEXT item's batch (u < 4: 72 registers) is requested
    // behind this item's fill and lands under its transforms and read-out -- the radix-16 passes (134 registers) leave room for half a
    // batch, not for a whole one; the second half goes out at the start of the item and lands under the fill of the first.  (An item
    // used to wait 21k of its 52k cycles for its one batch with nothing to cover it: one work-group per CU.)
    struct __attribute__((aligned(8))) Pair { double x, y; };
    double2 z[RING_FB][4];
    Pair wsy[RING_FB / 2];
    const bool pw = MODE == 0 && !WSYM && pixw != nullptr;
    constexpr bool wsym = MODE == 0 && WSYM;
    auto request = [&](auto U0C, const RingDesc &d, int c) __attribute__((always_inline)) {
        constexpr int U0 = decltype(U0C)::value;
        const bool hS = d.sS >= 0;
        const double *mpN = maps + (long long)c * P.npix + d.sN, *mpS = maps + (long long)c * P.npix + (hS ? d.sS : d.sN);
        const double *pwN = pixw + d.sN, *pwS = pixw + (hS ? d.sS : d.sN);
        const double2 *zp = zin + (long long)c * P.ny + d.sN;
#pragma unroll
        for (int u = U0; u < U0 + RING_FB / 2; u += 2) {
            const int j = 2 * (tid + (u >> 1) * nt), jj = j + 1 < n ? j : 0;
            wsy[u >> 1] = Pair{1.0, 1.0};
            if (MODE == 0 && wsym) wsy[u >> 1] = *reinterpret_cast<const Pair *>(pwN + jj);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = jj + q * n;
                if (MODE == 0) {
                    Pair fn = *reinterpret_cast<const Pair *>(mpN + i), fs = *reinterpret_cast<const Pair *>(mpS + i);
                    if (!wsym && pw) {  // (generic weight arrays: applied at the load -- their raw values would need another 128 registers)
                        const Pair wn = *reinterpret_cast<const Pair *>(pwN + i), ws = *reinterpret_cast<const Pair *>(pwS + i);
                        fn.x *= wn.x; fn.y *= wn.y; fs.x *= ws.x; fs.y *= ws.y;
                    }
                    z[u][q] = make_double2(fn.x, fn.y);
                    z[u + 1][q] = make_double2(fs.x, fs.y);
                } else {
                    z[u][q] = zp[i];
                    z[u + 1][q] = zp[i + 1];
                }
            }
        }
    };
    auto finish = [&](bool hS) __attribute__((always_inline)) {
        if (MODE != 0) return;
#pragma unroll
        for (int u = 0; u < RING_FB; u += 2) {
            const Pair w = wsy[u >> 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 rn = z[u][q], rs = z[u + 1][q];
                const double nx = wsym ? rn.x * w.x : rn.x, ny = wsym ? rn.y * w.y : rn.y;
                const double sx = wsym ? rs.x * w.x : rs.x, sy = wsym ? rs.y * w.y : rs.y;
                z[u][q] = make_double2(nx, hS ? sx : 0.0);
                z[u + 1][q] = make_double2(ny, hS ? sy : 0.0);
            }
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, RING_FB / 2>;
    bool have_half = false;  // the first half of this item's batch was requested by the item before it
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        // one round per item: item = 8 (2 s + rpair) + x -- the two items of set s = (ring set s / nb, component s % nb) on XCD x
        const int rpair = (item >> 3) & 1;  // round 0 = sub-DFTs 0 and 2, round 1 = sub-DFTs 1 and 3
        const int ring = ring_of(item), c = comp_of(item);
        const int itn = item + gridDim.x, ringn = itn < nitems ? ring_of(itn) : nrings;
        const bool nextv = ringn < nrings;
        if (ring >= nrings) {
            if (nextv) cur = desc[ringn];
            have_half = false;
            continue;
        }
        const long long sN = cur.sN, sS = cur.sS;
        asm volatile("; item" : "+v"(tid));
        const bool haveS = sS >= 0;
        if (!have_half) request(H0{}, cur, c);
        request(H1{}, cur, c);
        int4 nd0 = make_int4(0, 0, 0, 0);
        if (nextv) nd0 = *(reinterpret_cast<const int4 *>(desc + ringn) + (tid >> 30));
        {
        // (the previous round's read-out has to be over before the buffers are filled again)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        finish(haveS);
#pragma unroll
        for (int u = 0; u < RING_FB; ++u) {
            const int j = 2 * (tid + (u >> 1) * nt) + (u & 1);
            if (j >= n) continue;
            const double2 e0 = rpair ? csub(z[u][0], z[u][2]) : cadd(z[u][0], z[u][2]);
            const double2 e1 = rpair ? mul_mi(csub(z[u][1], z[u][3])) : cadd(z[u][1], z[u][3]);
            // load phases exp(-i pi j r / 2n) of r = rpair and r + 2: j < n and r <= 3, so j r < 4n needs no reduction, and r = 0 no phase at all
            const double2 t0 = cadd(e0, e1);
            buf[lds_slot(j)] = rpair ? cmul(t0, phase((unsigned)j)) : t0;
            buf[MP + lds_slot(j)] = cmul(csub(e0, e1), phase((unsigned)j * (unsigned)(rpair + 2)));
        }
        __syncthreads();
        // the next item's record: taken HERE, in front of this item's stores -- waited for behind them (their number is not known to
        // the compiler) it is s_waitcnt vmcnt(0), and the next item's loads are issued when the last store has been acknowledged
        have_half = false;
        if (nextv) {
            cur.sN = ((long long)__builtin_amdgcn_readfirstlane(nd0.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(nd0.x);
            cur.sS = ((long long)__builtin_amdgcn_readfirstlane(nd0.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(nd0.z);
            request(H0{}, cur, comp_of(itn));  // (z is free behind the last fill of the item)
            have_half = true;
        }
        const int half = tid >= nh ? 1 : 0, gt = tid - half * nh;
        double2 *bh = buf + half * MP;
        lds_fft_dif(bh, M, twf, P.twN, gt, nh);
        const int r = rpair + 2 * half;
        if (MODE == 1 && pixout) {  // straight to the pixels 4 k + r of the two rings, or to the residual (see k_ring_subdft)
            double *pxN = pixout + (long long)c * P.npix + sN + r, *pxS = pixout + (long long)c * P.npix + (haveS ? sS : sN) + r;
            if (ref) {
                const double *rfN = ref + (long long)c * P.npix + sN + r, *rfS = ref + (long long)c * P.npix + (haveS ? sS : sN) + r;
                for (int k = gt; k < n; k += nh) {
                    const double2 v = bh[lds_slot(bitrev(k, p))];
                    pxN[4 * k] = rfN[4 * k] - v.x;
                    if (haveS) pxS[4 * k] = rfS[4 * k] + v.y;
                }
            } else {
                for (int k = gt; k < n; k += nh) {
                    const double2 v = bh[lds_slot(bitrev(k, p))];
                    pxN[4 * k] = v.x;
                    if (haveS) pxS[4 * k] = -v.y;
                }
            }
        } else {
            double2 *out = Y + (long long)c * P.ny + sN + (long long)r * n;
            // (MODE 0: bins between lmax and 4n - lmax are not written, see k_ring_subdft)
            const int kdrop0 = MODE == 0 ? (P.lmax - r) / 4 + 1 : n, kdrop1 = (4 * n - P.lmax - r + 3) / 4;
            for (int k = gt; k < n; k += nh)
                if (k < kdrop0 || k >= kdrop1) *(out + k) = bh[lds_slot(bitrev(k, p))];
        }
        }
    }
}

// (Round 6, measured and not kept: the same work item with its two transforms run one after the other through ONE buffer by half as many threads, so
// that two independent groups fit a CU -- analysis 51.1 -> 50.6 ms per step, synthesis of ten fields 51.4 -> 56.1 ms: the item is bound by the issue of its
// vector instructions and by dependent LDS round trips, not by latency a second group could cover.  profiles/r06_pairseq_experiment.txt.)
// =====================================================================================
// 1b. rings whose Bluestein convolution does not fit LDS (nside 8192: cap rings with 4096 < n < 8192 need M = 16384 points
//     = 256 KiB): the length-M cyclic convolution as an EVEN and an ODD half of C = M / 2 points each
//         X[2k]   = FFT_C( x[j] + x[j + C] )[k],      X[2k+1] = FFT_C( (x[j] - x[j + C]) W_M^j )[k]         (forward, DIF)
//         y[j]    = IFFT_C(Y_even)[j] + W_M^-j IFFT_C(Y_odd)[j],   j < C                                      (inverse, DIT)
//     The input has n <= C non-zero points (x[j + C] = 0) and only y[0..n) is wanted, so each half is exactly the in-LDS
//     pipeline of the other rings (FFT_C -> filter -> IFFT_C) on one C-point buffer; the even half's result waits in
//     registers while the odd half runs.  The filter spectra are stored as [even bins | odd bins].
// =====================================================================================
constexpr int SPLIT_JMAX = 16;  // values of j per thread of the split kernels (n <= C = 16 x 512 threads at most; 512 threads: 256 registers for the radix-16 passes)

__global__ __launch_bounds__(512) void k_init_bhat_split(PlanDev P, const int *__restrict__ rp_list, int C,
                                                          double2 *__restrict__ bhat)
{
    extern __shared__ double2 buf[];
    const int rp = rp_list[blockIdx.x];
    const int n = P.nsub[rp], M = 2 * C;
    double2 *out = bhat + P.bhat_off[rp];
    // filter b[j] = chirp(j) for j < n, b[M - j] = chirp(j), 0 elsewhere; halves b0 = b[0..C), b1 = b[C..M)
    auto b_at = [&](int j) {  // 0 <= j < M
        const int jj = j < n ? j : (M - j < n ? M - j : -1);
        return jj < 0 ? make_double2(0.0, 0.0) : expipi((double)chirp_num(jj, n) / (double)n);
    };
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        for (int j = threadIdx.x; j < C; j += blockDim.x) {
            const double2 b0 = b_at(j), b1 = b_at(j + C);
            buf[lds_slot(j)] = half == 0 ? cadd(b0, b1) : cmul(csub(b0, b1), P.tw[j]);  // tw[j] = W_M^j (twN = M)
        }
        __syncthreads();
        lds_fft_dif(buf, C, P.tw, P.twN);
        for (int j = threadIdx.x; j < C; j += blockDim.x) out[half * C + j] = buf[lds_slot(j)];
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_ring_subdft_split(PlanDev P, const int *__restrict__ rp_list, int C,
                                                            const double *__restrict__ maps, const double *__restrict__ pixw,
                                                            const double2 *__restrict__ zin, double2 *__restrict__ Y,
                                                            double *__restrict__ pixout = nullptr, const double *__restrict__ ref = nullptr)
{
    extern __shared__ double2 buf[];
    const int rp = rp_list[blockIdx.y];
    const int c = blockIdx.z;
    const int n = P.nsub[rp];
    const long long sN = P.startN[rp], sS = P.startS[rp];
    const int M = 2 * C;
    const bool plain = fft_size_for(n) == n;  // a power of two (n == C): one plain FFT, no convolution
    const double *mp = maps + (long long)c * P.npix;
    const double2 *zp = zin + (long long)c * P.ny + sN;
    const double2 *bh = P.bhat + (plain ? 0 : P.bhat_off[rp]);
    const double inv = 1.0 / M;
    __shared__ double2 tw_hi[TW_HI_MAX], tw_lo[64];
    const TwFactored twf = load_tw_factored(tw_hi, tw_lo, P.tw, P.twN);
    auto z_at = [&](int j, int q) {  // packed ring value z_q[j] = z[j + q n]
        if (MODE == 0) {
            const long long iN = sN + j + (long long)q * n;
            double fn = mp[iN];
            if (pixw) fn *= pixw[iN];
            double fs = 0.0;
            if (sS >= 0) {
                const long long iS = sS + j + (long long)q * n;
                fs = mp[iS];
                if (pixw) fs *= pixw[iS];
            }
            return make_double2(fn, fs);
        }
        return zp[j + (long long)q * n];
    };
    // Nothing is carried in registers across the transforms (round 6): until then the 16 input values a0[j] and the 16 results of the even
    // half sat in registers under the radix-16 passes -- 256 registers, 1 050 more in scratch (3.2 KB per lane), 35 ms for the cap rings of
    // two maps at nside 8192, two thirds of their ring stage.  Now the input of the odd half is formed again from the pixels (L2) and the
    // even half's result waits where it belongs -- in Y, or in the pixels -- for the odd half to be ADDED to it by the same thread.
    // (MODE 1 with pixout: bin k of sub-DFT r goes straight to the pixel pair 4 k + r of the two rings, see k_ring_subdft.)
    auto emit = [&](int r, int k, double2 v, bool add) __attribute__((always_inline)) {
        if (MODE == 1 && pixout) {
            const long long iN = (long long)c * P.npix + sN + 4 * k + r, iS = (long long)c * P.npix + sS + 4 * k + r;
            if (add) {  // (what is there is ref - first part, or the first part: subtract / add the second)
                pixout[iN] += ref ? -v.x : v.x;
                if (sS >= 0) pixout[iS] += ref ? v.y : -v.y;
            } else {
                double fn = v.x, fs = -v.y;
                if (ref) { fn = ref[iN] - fn; if (sS >= 0) fs = ref[iS] - fs; }
                pixout[iN] = fn;
                if (sS >= 0) pixout[iS] = fs;
            }
        } else {
            double2 *o = Y + (long long)c * P.ny + sN + (long long)r * n + k;
            *o = add ? cadd(*o, v) : v;
        }
    };
    // t_r[j] exp(-i pi (j r + 2 j^2 [Bluestein]) / 2n)
    auto input = [&](int r, int j) __attribute__((always_inline)) {
        const double2 t = dif4_combine(z_at(j, 0), z_at(j, 1), z_at(j, 2), z_at(j, 3), r);
        const unsigned qn = load_phase_num(j, r, n, !plain);
        return qn ? cmul(t, expipi(-(double)qn / (2.0 * n))) : t;
    };
    for (int r = 0; r < 4; ++r) {
        if (plain) {
            __syncthreads();
#pragma unroll 4
            for (int u = 0; u < SPLIT_JMAX; ++u) {
                const int j = threadIdx.x + u * blockDim.x;
                if (j < n) buf[lds_slot(j)] = input(r, j);
            }
            __syncthreads();
            lds_fft_dif(buf, n, twf, P.twN);
            const int pbits = ilog2(n);
            for (int k = threadIdx.x; k < n; k += blockDim.x) emit(r, k, buf[lds_slot(bitrev(k, pbits))], false);
            continue;
        }
        for (int half = 0; half < 2; ++half) {
            __syncthreads();  // the previous pass has been read out of buf
#pragma unroll 4
            for (int u = 0; u < SPLIT_JMAX; ++u) {
                const int j = threadIdx.x + u * blockDim.x;
                if (j < C) {
                    double2 v = make_double2(0.0, 0.0);
                    if (j < n) {
                        v = input(r, j);
                        if (half) v = cmul(v, twf[j]);
                    }
                    buf[lds_slot(j)] = v;
                }
            }
            __syncthreads();
            lds_fft_dif(buf, C, twf, P.twN);
            for (int j = threadIdx.x; j < C; j += blockDim.x) buf[lds_slot(j)] = cmul(buf[lds_slot(j)], bh[half * C + j]);
            __syncthreads();
            lds_fft_dit_inv(buf, C, twf, P.twN);
#pragma unroll 4
            for (int u = 0; u < SPLIT_JMAX; ++u) {
                const int k = threadIdx.x + u * blockDim.x;
                if (k < n) {
                    double2 y = buf[lds_slot(k)];
                    if (half) y = cmulc(y, twf[k]);  // W_M^-k y_odd[k]
                    emit(r, k, cscale(cmul(y, expipi(-(double)chirp_num(k, n) / (double)n)), inv), half != 0);
                }
            }
        }
    }
}

// =====================================================================================
// synthesis: Fv -> ring spectra -> pixels (the Legendre part lives in hx_legendre_valu.hip)
// =====================================================================================
// Fv[m][rp][4 nc] -> conj(Z) spectra of the packed ring pair z = f_N + i f_S.  Fv is the output of the vector-unit synthesis
// (hx_legendre_valu.hip) for ONE map (nc = 1) or (Q, U) field (nc = 2), component c = (N_re, N_im, S_re, S_im) at 4 c.
// X[k] = sum_{m == k mod nphi} (c_m/2) Ft_m + sum_{m == -k} (c_m/2) conj(Ft_m), Ft = F e^{i m phi0};  output
// Zc[c][startN + k] = conj(X_N + i X_S).  The values of four consecutive ring pairs at one m share a 128-byte line: a group
// takes four ring pairs (thread = (ring pair, component, k)).
__global__ __launch_bounds__(256) void k_synth_spectrum_v(PlanDev P, const double *__restrict__ Fv, int nc, int lmax,
                                                          double2 *__restrict__ Zc, const int *__restrict__ mlim, int rp_hi)
{
    const int rp = blockIdx.x * 4 + (threadIdx.x & 3), rest = threadIdx.x >> 2;
    if (rp >= P.nrp || rp >= rp_hi) return;  // (ring pairs from rp_hi on: k_synth_spectrum_t)
    // nc <= 64 components side by side (nc need not divide 64: the threads left over have nothing to do)
    const int c = rest % nc, kk = rest / nc, kstep = (int)(blockDim.x >> 2) / nc;
    if (kk >= kstep) return;
    const int n = P.nsub[rp], nphi = 4 * n;
    const bool shifted = P.shifted[rp] != 0;
    const long long mstride = (long long)P.nrp_pad * 4 * nc;
    const double *row = Fv + (long long)rp * 4 * nc + 4 * c;
    // mlim (batched matrix-unit synthesis): rows of this ring pair exist for m <= mlim[rp] only (pruned beyond: zero, and not written)
    const int mtop = mlim ? min(lmax, mlim[rp]) : lmax;
    // One pass serves the bins k and nphi - k: A = sum_{m == k} (c_m / 2) Ft_m, B = sum_{m == -k} (c_m / 2) Ft_m;
    // X[k] = A + conj(B), X[nphi - k] = B + conj(A) -- every value of Fv is read once (round 5; two passes before)
    for (int k = kk; 2 * k <= nphi; k += kstep) {
        const int k2 = (nphi - k) % nphi;
        double2 an = make_double2(0.0, 0.0), as = an, bn = an, bs = an;
        for (int m = k; m <= mtop; m += nphi) {  // m == k (mod nphi)
            const double2 *b = reinterpret_cast<const double2 *>(row + m * mstride);
            double2 ph = make_double2(1.0, 0.0);
            if (shifted) ph = expipi((double)(m % (2 * nphi)) / (double)nphi);
            const double sc = m == 0 ? 0.5 : 1.0;  // c_m / 2
            an = cadd(an, cscale(cmul(b[0], ph), sc));
            as = cadd(as, cscale(cmul(b[1], ph), sc));
        }
        if (k2 != k) {
            for (int m = k2; m <= mtop; m += nphi) {  // m == -k (mod nphi)
                const double2 *b = reinterpret_cast<const double2 *>(row + m * mstride);
                double2 ph = make_double2(1.0, 0.0);
                if (shifted) ph = expipi((double)(m % (2 * nphi)) / (double)nphi);
                const double sc = m == 0 ? 0.5 : 1.0;
                bn = cadd(bn, cscale(cmul(b[0], ph), sc));
                bs = cadd(bs, cscale(cmul(b[1], ph), sc));
            }
        } else {
            bn = an;
            bs = as;
        }
        double2 *z = Zc + (long long)c * P.ny + P.startN[rp];
        z[k] = cconj(cadd(cadd(an, cconj(bn)), mul_pi(cadd(as, cconj(bs)))));
        if (k2 != k) z[k2] = cconj(cadd(cadd(bn, cconj(an)), mul_pi(cadd(bs, cconj(as)))));
    }
}

// The same for ring pairs with 4 n >= 2 lmax + 2 pixels per ring (round 5; at nside 4096 / lmax 6144: 5120 of the 8192 ring pairs, 73 % of
// the pixels): no two orders fall on one bin, so the pass is a TRANSPOSITION -- bin m = conj(Ft_N + i Ft_S), bin nphi - m =
// conj(conj Ft_N + i conj Ft_S), zeros between lmax and nphi - lmax -- and goes through LDS: a block takes one ring pair and 64
// orders, reads their rows of Fv (nc x 32 contiguous bytes each), and writes, per component, two runs of 64 consecutive bins.  The
// gather above walks the orders per bin with one or two 32-byte reads in flight per thread and scatters 16-byte writes: 48 ms for the
// twenty components of ten fields against ~16 ms of traffic at copy rate.
constexpr int SPT_M = 64;
__global__ __launch_bounds__(256) void k_synth_spectrum_t(PlanDev P, const double *__restrict__ Fv, int nc, int lmax, double2 *__restrict__ Zc,
                                                          const int *__restrict__ mlim, int rp_lo)
{
    extern __shared__ double2 spt[];  // [2][nc][SPT_M]: the bins m and nphi - m of the block's orders
    const int rp = rp_lo + blockIdx.x;
    const int n = P.nsub[rp], nphi = 4 * n, m0 = blockIdx.y * SPT_M;
    if (2 * m0 > nphi) return;  // (orders beyond nphi / 2 belong to the mirrored run of another block)
    const bool shifted = P.shifted[rp] != 0;
    const int mtop = mlim ? min(lmax, mlim[rp]) : lmax;
    const long long mstride = (long long)P.nrp_pad * 4 * nc;
    const double *row = Fv + (long long)rp * 4 * nc;
    // element e = (order j of the tile, component c): 32 contiguous bytes; consecutive threads take consecutive components of an order
    for (int e = threadIdx.x; e < SPT_M * nc; e += blockDim.x) {
        const int j = e / nc, c = e % nc, m = m0 + j;
        double2 z1 = make_double2(0.0, 0.0), z2 = z1;
        if (m <= mtop) {
            const double2 *b = reinterpret_cast<const double2 *>(row + m * mstride + 4 * c);
            double2 fn = b[0], fs = b[1];
            if (shifted) {
                const double2 ph = expipi((double)(m % (2 * nphi)) / (double)nphi);
                fn = cmul(fn, ph);
                fs = cmul(fs, ph);
            }
            if (m == 0) {  // c_0 / 2 = 1 / 2 and both sums meet in bin 0
                z1 = cconj(cadd(make_double2(fn.x, 0.0), mul_pi(make_double2(fs.x, 0.0))));
            } else {
                z1 = cconj(cadd(fn, mul_pi(fs)));
                z2 = cconj(cadd(cconj(fn), mul_pi(cconj(fs))));
            }
        }
        spt[c * SPT_M + j] = z1;
        spt[(nc + c) * SPT_M + j] = z2;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SPT_M * nc; e += blockDim.x) {
        const int c = e / SPT_M, j = e % SPT_M, m = m0 + j;
        double2 *z = Zc + (long long)c * P.ny + P.startN[rp];
        if (2 * m <= nphi) z[m] = spt[c * SPT_M + j];
        if (m > 0 && 2 * m < nphi) z[nphi - m] = spt[(nc + c) * SPT_M + j];
    }
}

// (Until round 6 the inverse sub-DFTs wrote their spectra to Y and a pass of its own, k_synth_scatter, turned Y_r[k] = conj(z[4k + r]) into
// pixels: 20 ms per ten fields.  The read-out of the sub-DFT kernels writes the pixels itself now -- MODE 1 with `pixout`: ten fields
// 62.7 -> 52.7 ms of ring stage, ten maps 31.0 -> 25.9, same device -- and a synthesis holds no Y buffer.)
}  // namespace hx

using namespace hx;

// =====================================================================================
// plan
// =====================================================================================
PlanDev hx_plan::dev() const
{
    PlanDev P;
    P.nside = nside; P.lmax = lmax; P.nrp = nrp; P.nrp_pad = nrp_pad; P.twN = twN;
    P.npix = npix; P.ny = ny;
    P.z = z.as<double>(); P.omz = omz.as<double>(); P.sth = sth.as<double>(); P.rwdef = rwdef.as<double>();
    P.nsub = nsub.as<int>(); P.shifted = shifted.as<int>();
    P.startN = startN.as<long long>(); P.startS = startS.as<long long>(); P.bhat_off = bhat_off.as<long long>();
    P.tw = tw.as<double2>(); P.bhat = bhat.as<double2>();
    P.mfac = mfac.as<double>(); P.kfac2 = kfac2.as<double>();
    P.rec0 = nullptr; P.rec2 = nullptr;
    P.wnorm = wnorm; P.hsrc = hsrc; P.hsrc_stride = hsrc_stride; P.hN = eqN;
    P.nssrc = nssrc; P.ns_m0 = ns_m0; P.ns_ms = m_step;
    return P;
}

// Longest FFT done in LDS (points; 8192 = 128 KiB of the CU's 160).  Lowering it (hx_set_max_lds_fft, a power of two >= 16)
// sends smaller rings through the split kernels -- the way the tests exercise them without an nside-8192 map.
static int g_lds_fft_cap = 8192;
static int lds_fft_cap() { return g_lds_fft_cap; }
extern "C" int hx_set_max_lds_fft(int points)
{
    if (points < 16 || points > 8192 || (points & (points - 1))) return fail(HX_ERR_ARG, "hx_set_max_lds_fft: a power of two in [16, 8192]");
    g_lds_fft_cap = points;
    return HX_OK;
}

// Tables that depend on the band limit only (twiddles of the in-LDS FFT, seeds and coefficients of the recursions) and the
// kernel attributes: shared by the HEALPix plan and the equiangular plan of the point transform (hx_nufft.hip).
static int plan_tables(hx_plan *pl)
{
    const int lmax = pl->lmax;
    std::vector<double2> tw(std::max(pl->twN / 2, 64));  // load_tw_factored copies 64 entries whatever twN
    for (int k = 0; k < (int)tw.size(); ++k) {
        long double a = -2.0L * 3.141592653589793238462643383279502884L * k / pl->twN;
        tw[k].x = (double)cosl(a); tw[k].y = (double)sinl(a);
    }
    // mfac[m] = (-1)^m sqrt((2m+1)/(4pi) prod_{k<=m} (2k-1)/(2k));  kfac2[m] = K_m 2^-(m-2)
    std::vector<double> mfac(lmax + 1), kfac2(lmax + 3, 0.0);
    {
        long double p = 1.0L;
        for (int m = 0; m <= lmax; ++m) {
            if (m > 0) p *= (2.0L * m - 1.0L) / (2.0L * m);
            long double v = sqrtl((2.0L * m + 1.0L) / (4.0L * 3.141592653589793238462643383279502884L) * p);
            mfac[m] = (double)((m & 1) ? -v : v);
        }
        long double k = 1.0L;
        for (int m = 2; m <= lmax + 2; ++m) {
            if (m > 2) k *= sqrtl((2.0L * m) * (2.0L * m - 1.0L) / ((m - 2.0L) * (m + 2.0L))) / 2.0L;
            kfac2[m] = (double)k;
        }
    }
    HX_TRY(upload(pl->tw, tw));
    HX_TRY(upload(pl->mfac, mfac));
    HX_TRY(upload(pl->kfac2, kfac2));
    hipStream_t st = rt().stream;
    HX_TRY(pl->cn0.alloc(sizeof(double2) * (pl->nlm + TABLE_PAD)));
    HX_TRY(pl->al0.alloc(sizeof(double) * (pl->nlm + TABLE_PAD)));
    HX_HIP(hipMemsetAsync(pl->cn0.p, 0, sizeof(double2) * (pl->nlm + TABLE_PAD), st));
    HX_HIP(hipMemsetAsync(pl->al0.p, 0, sizeof(double) * (pl->nlm + TABLE_PAD), st));
    hipLaunchKernelGGL(k_init_norm0, dim3((2 * (lmax + 1) + 63) / 64), dim3(64), 0, st, lmax, pl->cn0.as<double2>(), pl->al0.as<double>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_init_bhat), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // 3 KiB of the 160 KiB are the static twiddle tables of k_ring_subdft
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_pairfft<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_pairfft<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_pairfft<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_init_bhat_split), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft_split<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft_split<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    HX_HIP(hipGetLastError());
    return HX_OK;
}

extern "C" hx_plan *hx_plan_create(int nside, int lmax, int max_comp)
{
    if (ensure_ready() != HX_OK) return nullptr;
    if (nside < 1 || lmax < 0 || max_comp < 1) {
        set_error("hx_plan_create: bad argument");
        return nullptr;
    }
    hx_plan *pl = new hx_plan;
    pl->nside = nside; pl->lmax = lmax; pl->max_comp = max_comp;
    pl->npix = 12LL * nside * nside;
    pl->nrp = 2 * nside;
    pl->nrp_pad = (pl->nrp + 63) / 64 * 64;
    pl->nlm = (long long)(lmax + 1) * (lmax + 2) / 2;
    pl->wnorm = 4.0 * M_PI / (double)pl->npix;
    const long long ns = nside, ncap = 2 * ns * (ns - 1);
    std::vector<double> z(pl->nrp), omz(pl->nrp), sth(pl->nrp), rw(pl->nrp, 1.0);
    std::vector<int> nsub(pl->nrp), shifted(pl->nrp);
    std::vector<long long> sN(pl->nrp), sS(pl->nrp), boff(pl->nrp, -1);
    const double fact2 = 4.0 / (double)pl->npix, fact1 = (double)(2 * ns) * fact2;
    int maxM = 1;
    for (int rp = 0; rp < pl->nrp; ++rp) {
        const int i = rp + 1;
        if (i < nside) {
            double tmp = (double)i * (double)i * fact2;
            z[rp] = 1.0 - tmp; omz[rp] = tmp; sth[rp] = sqrt(tmp * (2.0 - tmp));
            nsub[rp] = i; sN[rp] = 2LL * i * (i - 1); shifted[rp] = 1;
        } else {
            z[rp] = (double)(2 * nside - i) * fact1; omz[rp] = 1.0 - z[rp];
            sth[rp] = sqrt((1.0 - z[rp]) * (1.0 + z[rp]));
            nsub[rp] = nside; sN[rp] = ncap + (long long)(i - nside) * 4 * ns;
            shifted[rp] = ((i - nside) & 1) == 0;
        }
        sS[rp] = i == 2 * nside ? -1 : pl->npix - sN[rp] - 4LL * nsub[rp];
        maxM = std::max(maxM, fft_size_for(nsub[rp]));
    }
    pl->ny = sN[pl->nrp - 1] + 4LL * nsub[pl->nrp - 1];
    // in-LDS FFT length limit (8192 points = 128 KiB); Bluestein rings of twice that run as two halves (split kernels)
    const int cap = lds_fft_cap();
    if (maxM > 2 * cap || (maxM > cap && nside > 0 && [&] { for (int rp = 0; rp < pl->nrp; ++rp) if (fft_size_for(nsub[rp]) > cap && fft_size_for(nsub[rp]) == nsub[rp]) return true; return false; }())) {
        set_error("hx_plan_create: nside=%d needs an in-LDS FFT of %d points (limit %d, %d for Bluestein rings); unsupported", nside, maxM, cap, 2 * cap);
        delete pl;
        return nullptr;
    }
    pl->fft_cap = cap;
    {
        // FFT-size classes: ring pairs grouped by in-LDS FFT length, longest rings first; rings beyond the limit form
        // the split class (M = 2 x cap)
        // (second key: 0 regular kernel; -1 plain 2^k rings whose two buffers fit LDS: the pair kernel)
        std::map<std::pair<int, int>, std::vector<int>> byM;
        for (int rp = pl->nrp - 1; rp >= 0; --rp) {
            const int M = fft_size_for(nsub[rp]);
            const bool pair = M == nsub[rp] && M >= 16 && M <= cap && (2 * (size_t)lds_fft_slots(M) + ring_ph_hi(M) + 64) * sizeof(double2) + 4096 <= 160 * 1024;
            byM[{M, pair ? -1 : 0}].push_back(rp);
        }
        std::vector<int> list;
        for (auto it = byM.rbegin(); it != byM.rend(); ++it) {
            hx_plan::FftClass c;
            c.M = it->first.first; c.big = it->first.second; c.first = (int)list.size(); c.count = (int)it->second.size();
            list.insert(list.end(), it->second.begin(), it->second.end());
            pl->fft_classes.push_back(c);
        }
        if (upload(pl->fft_rp_list, list) != HX_OK) { delete pl; return nullptr; }
        pl->h_fft_rp_list = list;
    }
    pl->twN = std::max(maxM, 2);
    pl->lds_fft = (size_t)lds_fft_slots(std::min(maxM, cap)) * sizeof(double2);
    pl->h_sth = sth; pl->h_z = z; pl->h_nsub = nsub; pl->h_startN = sN; pl->h_startS = sS;
    // Bluestein tables: one spectrum per distinct non-power-of-two sub-length
    std::vector<int> blu_list, blu_split;
    long long btot = 0;
    {
        std::map<int, long long> off_of_n;
        for (int rp = 0; rp < pl->nrp; ++rp) {
            int n = nsub[rp], M = fft_size_for(n);
            if (M == n) continue;
            auto it = off_of_n.find(n);
            if (it == off_of_n.end()) {
                it = off_of_n.emplace(n, btot).first;
                btot += M;
                (M > cap ? blu_split : blu_list).push_back(rp);
            }
            boff[rp] = it->second;
        }
    }
    int rc = HX_OK;
    auto chk = [&](int r) { if (rc == HX_OK) rc = r; };
    chk(upload(pl->z, z)); chk(upload(pl->omz, omz)); chk(upload(pl->sth, sth)); chk(upload(pl->rwdef, rw));
    chk(upload(pl->nsub, nsub)); chk(upload(pl->shifted, shifted));
    chk(upload(pl->startN, sN)); chk(upload(pl->startS, sS)); chk(upload(pl->bhat_off, boff));
    {
        std::vector<RingDesc> desc(pl->h_fft_rp_list.size());
        for (size_t k = 0; k < desc.size(); ++k) {
            const int rp = pl->h_fft_rp_list[k];
            desc[k] = RingDesc{sN[rp], sS[rp], boff[rp], nsub[rp], rp};
        }
        chk(upload(pl->fft_desc, desc));
    }
    chk(pl->bhat.alloc(sizeof(double2) * std::max<long long>(btot, 1)));
    if (rc != HX_OK || plan_tables(pl) != HX_OK) { delete pl; return nullptr; }
    hipStream_t st = rt().stream;
    if (!blu_list.empty()) {
        DevBuf d_list;
        if (upload(d_list, blu_list) != HX_OK) { delete pl; return nullptr; }
        hipLaunchKernelGGL(k_init_bhat, dim3((unsigned)blu_list.size()), dim3(512), pl->lds_fft, st, pl->dev(),
                           d_list.as<int>(), pl->bhat.as<double2>());
        (void)hipStreamSynchronize(st);
    }
    if (!blu_split.empty()) {
        DevBuf d_list;
        if (upload(d_list, blu_split) != HX_OK) { delete pl; return nullptr; }
        hipLaunchKernelGGL(k_init_bhat_split, dim3((unsigned)blu_split.size()), dim3(std::min(512, std::max(64, cap / 16))), (size_t)lds_fft_slots(cap) * sizeof(double2), st,
                           pl->dev(), d_list.as<int>(), cap, pl->bhat.as<double2>());
        (void)hipStreamSynchronize(st);
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("hx_plan_create: table initialisation failed");
        delete pl;
        return nullptr;
    }
    return pl;
}

// Plan of the Legendre stages on N / 2 equidistant rings theta_j = 2 pi (j + 1/2) / N, j < N / 2 (N a multiple of 4): ring pair
// r < N / 4 = (theta_r, pi - theta_r), pole -> equator like the HEALPix pairs.  It has no pixels: its ring spectra h_m(theta_j)
// come from the non-uniform Fourier stage of the point transform (hx_nufft.hip), which owns the plan.
hx_plan *hx::plan_create_equiangular(int N, int lmax)
{
    if (N < 4 || (N & 3) || lmax < 0 || 2 * lmax + 1 >= N) {
        set_error("equiangular plan: N=%d lmax=%d", N, lmax);
        return nullptr;
    }
    hx_plan *pl = new hx_plan;
    pl->nside = 0; pl->lmax = lmax; pl->max_comp = 16;
    pl->npix = 0; pl->ny = 0;
    pl->eqN = N;
    pl->wnorm = 1.0 / N;
    pl->nrp = N / 4;
    pl->nrp_pad = (pl->nrp + 63) / 64 * 64;
    pl->nlm = (long long)(lmax + 1) * (lmax + 2) / 2;
    std::vector<double> z(pl->nrp), omz(pl->nrp), sth(pl->nrp), rw(pl->nrp, 1.0);
    for (int r = 0; r < pl->nrp; ++r) {
        const long double t = 2.0L * 3.141592653589793238462643383279502884L * (r + 0.5L) / N;
        const long double sh = sinl(0.5L * t);
        z[r] = (double)cosl(t); omz[r] = (double)(2.0L * sh * sh); sth[r] = (double)sinl(t);
    }
    pl->h_sth = sth; pl->h_z = z;
    pl->twN = 2;
    int rc = HX_OK;
    auto chk = [&](int r) { if (rc == HX_OK) rc = r; };
    chk(upload(pl->z, z)); chk(upload(pl->omz, omz)); chk(upload(pl->sth, sth)); chk(upload(pl->rwdef, rw));
    if (rc != HX_OK || plan_tables(pl) != HX_OK || hipStreamSynchronize(rt().stream) != hipSuccess) { delete pl; return nullptr; }
    return pl;
}

extern "C" void hx_plan_destroy(hx_plan *plan)
{
    if (!plan) return;
    if (rt().ready) (void)hipStreamSynchronize(rt().stream);
    for (int i = 0; i < hx_plan::NSTAGE; ++i) {
        if (plan->stage_done[i]) (void)hipEventDestroy(plan->stage_done[i]);
    }
    for (int i = 0; i < hx_plan::NUNIT_EV; ++i)
        if (plan->unit_up[i]) (void)hipEventDestroy(plan->unit_up[i]);
    delete plan;
}

extern "C" int hx_plan_release_scratch(hx_plan *pl)
{
    if (!pl) return fail(HX_ERR_ARG, "hx_plan_release_scratch: null plan");
    if (rt().ready) {
        HX_HIP(hipStreamSynchronize(rt().stream));
        if (rt().copy) HX_HIP(hipStreamSynchronize(rt().copy));
    }
    for (int i = 0; i < hx_plan::NSTAGE; ++i) pl->stage[i].release();
    pl->resid_maps.release();
    pl->Y.release();
    pl->F.release();
    pl->partial.release();
    pl->syn_tab.release();
    return HX_OK;
}

extern "C" int64_t hx_plan_scratch_bytes(const hx_plan *pl)
{
    if (!pl) return 0;
    return (int64_t)(pl->stage[0].bytes + pl->stage[1].bytes + pl->stage[2].bytes + pl->resid_maps.bytes + pl->Y.bytes + pl->F.bytes + pl->partial.bytes + pl->rec0.bytes + pl->rec2.bytes + pl->cn0.bytes + pl->al0.bytes + pl->cn2.bytes + pl->al2.bytes +
                     pl->bhat.bytes + pl->syn_tab.bytes);
}

extern "C" int hx_plan_last_chunks(const hx_plan *pl) { return pl ? pl->last_chunks : 0; }

namespace hx {
int ensure_rec2(hx_plan *pl)
{
    if (pl->cn2.p) return HX_OK;
    HX_TRY(pl->cn2.alloc(sizeof(double2) * (pl->nlm + TABLE_PAD)));
    HX_TRY(pl->al2.alloc(sizeof(double) * (pl->nlm + TABLE_PAD)));
    HX_HIP(hipMemsetAsync(pl->cn2.p, 0, sizeof(double2) * (pl->nlm + TABLE_PAD), rt().stream));
    HX_HIP(hipMemsetAsync(pl->al2.p, 0, sizeof(double) * (pl->nlm + TABLE_PAD), rt().stream));
    hipLaunchKernelGGL(k_init_norm2, dim3((pl->lmax + 64) / 64), dim3(64), 0, rt().stream, pl->lmax, pl->cn2.as<double2>(), pl->al2.as<double>());
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// One launch per FFT-size class, so that every class gets the LDS it needs and no more
// (a 4096-point ring must not reserve the 128 KiB of an 8192-point Bluestein ring).
template <int MODE>
static int launch_subdft_classes(hx_plan *pl, int nb, const double *d_maps, const double *d_pw, const double2 *zin, double2 *Y, int rp_lo = 0,
                                 int rp_hi = 0x7fffffff, double *pixout = nullptr, const double *ref = nullptr)
{
    for (const auto &cls : pl->fft_classes) {
        // the ring pairs of the class that lie in [rp_lo, rp_hi): its list is in descending order, so they are one run of it
        hx_plan::FftClass c = cls;
        if (rp_lo > 0 || rp_hi < pl->nrp) {
            const int *b = pl->h_fft_rp_list.data() + cls.first, *e = b + cls.count;
            const int *x = std::lower_bound(b, e, rp_hi, [](int rp, int lim) { return rp >= lim; });  // first rp < rp_hi
            const int *y = std::lower_bound(b, e, rp_lo, [](int rp, int lim) { return rp >= lim; });  // first rp < rp_lo
            c.first = cls.first + (int)(x - b);
            c.count = (int)(y - x);
            if (c.count <= 0) continue;
        }
        if (c.big < 0) {  // plain 2^k rings: two sub-DFTs per work item (k_ring_pairfft)
            const int threads = 2 * std::max(64, c.M / 16);
            const size_t lds = (size_t)(2 * lds_fft_slots(c.M) + ring_ph_hi(c.M) + 64) * sizeof(double2);
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(512 / threads, (160 * 1024) / (lds + 3 * 1024 + 256)));
            const long long items = ((long long)c.count + 7) / 8 * nb * 16;
            const unsigned groups = (unsigned)std::min<long long>(items, (long long)rt().cus * per_cu);
            if (MODE == 0 && d_pw && pl->pw_mode == 2) {
                hipLaunchKernelGGL((k_ring_pairfft<MODE, MODE == 0>), dim3(groups), dim3(threads), lds, rt().stream,
                                   pl->dev(), pl->fft_desc.as<RingDesc>() + c.first, c.count, nb, c.M, d_maps, d_pw, zin, Y, pixout, ref);
            } else {
                hipLaunchKernelGGL((k_ring_pairfft<MODE, false>), dim3(groups), dim3(threads), lds, rt().stream,
                                   pl->dev(), pl->fft_desc.as<RingDesc>() + c.first, c.count, nb, c.M, d_maps, d_pw, zin, Y, pixout, ref);
            }
            continue;
        }
        if (c.M > pl->fft_cap || c.big) {  // Bluestein convolution of 2 x cap points in two halves / plain FFT of > 4096 points
            const int C = std::min(c.M, pl->fft_cap), threads = std::min(512, std::max(64, C / SPLIT_JMAX));
            hipLaunchKernelGGL(k_ring_subdft_split<MODE>, dim3(1, c.count, nb), dim3(threads), (size_t)lds_fft_slots(C) * sizeof(double2), rt().stream,
                               pl->dev(), pl->fft_rp_list.as<int>() + c.first, C, d_maps, d_pw, zin, Y, pixout, ref);
            continue;
        }
        // M / 16 threads per group (one radix-16 butterfly per thread and pass); persistent groups, as many as the CUs hold at
        // once: LDS (the padded buffer, the phase tables behind it, 3 KiB of twiddle tables) and 8 waves per CU, so that every wave has 256 registers
        const int threads = std::min(RING_NTMAX, std::max(64, c.M / 16));
        const size_t lds = (size_t)(lds_fft_slots(c.M) + ring_ph_hi(c.M) + 64) * sizeof(double2);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(512 / threads, (160 * 1024) / (lds + 3 * 1024 + 256)));
        const long long items = ((long long)c.count + 7) / 8 * nb * 32;
        const unsigned groups = (unsigned)std::min<long long>(items, (long long)rt().cus * per_cu);
        if (MODE == 0 && d_pw && pl->pw_mode == 2) {
            hipLaunchKernelGGL((k_ring_subdft<MODE, MODE == 0>), dim3(groups), dim3(threads), lds, rt().stream,
                               pl->dev(), pl->fft_desc.as<RingDesc>() + c.first, c.count, nb, c.M, d_maps, d_pw, zin, Y, pixout, ref);
        } else {
            hipLaunchKernelGGL((k_ring_subdft<MODE, false>), dim3(groups), dim3(threads), lds, rt().stream,
                               pl->dev(), pl->fft_desc.as<RingDesc>() + c.first, c.count, nb, c.M, d_maps, d_pw, zin, Y, pixout, ref);
        }
    }
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// flag stays != 0 if the weights of every ring pair repeat over its four quadrants and from its northern to its southern ring
// (bitwise): one read of the array, 0.4 ms at nside 4096 per call, against 8 weight loads per pixel pair in every ring kernel
__global__ __launch_bounds__(256) void k_pixw_symmetry(PlanDev P, const double *__restrict__ pixw, int *__restrict__ flag)
{
    const int rp = blockIdx.x, n = P.nsub[rp];
    const long long sN = P.startN[rp], sS = P.startS[rp];
    bool ok = true;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double w = pixw[sN + j];
#pragma unroll
        for (int q = 1; q < 4; ++q) ok = ok && pixw[sN + j + (long long)q * n] == w;
        if (sS >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ok = ok && pixw[sS + j + (long long)q * n] == w;
        }
    }
    if (!ok) *flag = 0;
}

int launch_ring_subdft_maps(hx_plan *pl, int nb, const double *d_maps, const double *d_pw, double2 *Y, int rp_lo, int rp_hi)
{
    ProfScope ps("ring_fft");
    if (d_pw && pl->pw_checked != d_pw) HX_TRY(classify_pixel_weights(pl, d_pw));  // (entry points that did not do it themselves)
    return launch_subdft_classes<0>(pl, nb, d_maps, d_pw, nullptr, Y, rp_lo, rp_hi);
}

// Called by the entry points of the C ABI right after they have bound their pixel-weight array, before they queue anything else:
// the host waits for a 4-byte verdict (pl->pw_mode: 1 generic, 2 symmetric), which holds for this array until the next entry.
int classify_pixel_weights(hx_plan *pl, const double *d_pw)
{
    pl->pw_checked = d_pw;
    pl->pw_mode = 0;
    if (!d_pw || pl->nside < 1) return HX_OK;
    HX_TRY(pl->pw_sym.alloc(sizeof(int)));
    HX_HIP(hipMemsetAsync(pl->pw_sym.p, 1, sizeof(int), rt().stream));
    hipLaunchKernelGGL(k_pixw_symmetry, dim3(pl->nrp), dim3(256), 0, rt().stream, pl->dev(), d_pw, pl->pw_sym.as<int>());
    int flag = 0;
    HX_HIP(hipMemcpyAsync(&flag, pl->pw_sym.p, sizeof(int), hipMemcpyDeviceToHost, rt().stream));
    HX_HIP(hipStreamSynchronize(rt().stream));
    pl->pw_mode = flag != 0 ? 2 : 1;
    return HX_OK;
}
}  // namespace hx

// ---- one synthesis pass over a batch (device pointers): one sweep of the vector-unit kernel per map / field (the round-1 matrix
// kernel it replaces took 100 / 217 ms for one spin-0 map / spin-2 field at nside 4096 against 19 / 57, and 201 / 653 ms for ten
// against 187 / 570).  If d_ref != NULL the output is the residual ref - synth (Jacobi iteration). ----
// small batches: one sweep of the vector-unit kernel per four maps / two fields (they share the recursion), then the rest
static int synthesis_batch_valu(hx_plan *pl, int spin, int nb, const double2 *d_alms, double *d_maps, const double *d_ref)
{
    hipStream_t st = rt().stream;
    const int cpu = spin ? 2 : 1;
    PlanDev P = pl->dev();
    const int umax = synth_valu_max_units(spin);
    // ring modes and ring spectra live in the analysis' operand buffer F, as in the batched path (F is idle during a synthesis, and a plan
    // that has run a batched synthesis holds most of the HBM in it already: separate buffers failed to allocate at nside 8192)
    const size_t fv_pad = (sizeof(double) * (size_t)(pl->lmax + 1) * pl->nrp_pad * 4 * cpu * umax + 255) & ~(size_t)255;
    HX_TRY(pl->F.alloc(fv_pad + sizeof(double2) * (size_t)pl->ny * cpu * umax));
    double *fsyn = pl->F.as<double>();
    double2 *zc = reinterpret_cast<double2 *>(reinterpret_cast<char *>(pl->F.p) + fv_pad);
    for (int c0 = 0; c0 < nb;) {
        int units = umax;
        while (units * cpu > nb - c0) units >>= 1;
        const int nc = units * cpu;
        hx_plan::TaskSet *ts = nullptr;
        HX_TRY(valu_tasks(pl, spin, &ts, synth_valu_task_blocks(spin, units)));
        HX_TRY(launch_synth_valu(pl, spin, units, *ts, d_alms + (size_t)c0 * pl->nlm, fsyn));
        ProfScope ps("ring_fft");
        hipLaunchKernelGGL(k_synth_spectrum_v, dim3((pl->nrp + 3) / 4), dim3(256), 0, st, P, fsyn, nc, pl->lmax, zc, (const int *)nullptr, pl->nrp);
        // inverse sub-DFTs whose read-out writes the pixels (or the residual ref - synthesised of a Jacobi iteration) itself
        HX_TRY(launch_subdft_classes<1>(pl, nc, nullptr, nullptr, zc, nullptr, 0, 0x7fffffff, d_maps + (size_t)c0 * pl->npix,
                                        d_ref ? d_ref + (size_t)c0 * pl->npix : nullptr));
        c0 += nc;
    }
    HX_HIP(hipGetLastError());
    return HX_OK;
}

static int synthesis_batch(hx_plan *pl, int spin, int nb, const double2 *d_alms, double *d_maps,
                           const double *d_ref)
{
    hipStream_t st = rt().stream;
    const int cpu = spin ? 2 : 1;  // components per unit (map / field)
    PlanDev P = pl->dev();
    // batches of >= 5 maps / >= 3 fields: sweeps of up to 20 maps / 10 fields on the matrix unit (hx_synth_duo.hip).  Their ring
    // modes (Fv) and ring spectra (conj Z) live in the analysis' operand buffer F, which is idle during a synthesis: a Jacobi iteration
    // of ten fields needs no HBM beyond what its analysis passes hold (F 64 GB >= 32 + 32).
    const int nunits_all = nb / cpu;
    if (nunits_all >= (spin ? 3 : 5)) {
        const int umax = synth_duo_max_units(spin);
        hx_plan::TaskSet *ts = nullptr;
        HX_TRY(synth_duo_tasks(pl, spin, &ts));
        // what a sweep of `units` holds: ring modes + ring spectra (in F), the B-operand table
        auto sweep_bytes = [&](int units) {
            const double nc = (double)units * cpu;
            return sizeof(double) * (double)(pl->lmax + 1) * pl->nrp_pad * synth_duo_rowlen(spin, units) + sizeof(double2) * (double)pl->ny * nc +
                   (double)synth_duo_table_bytes(pl, spin, units);
        };
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
        double avail = 0.92 * ((double)fr + (double)pl->F.bytes + (double)pl->Y.bytes + (double)pl->syn_tab.bytes);
        if (scratch_budget_bytes() > 0.0) avail = std::min(avail, scratch_budget_bytes());  // (hx_set_scratch_budget bounds this scratch too)
        for (int u0 = 0; u0 < nunits_all;) {
            int units = std::min(umax, nunits_all - u0);
            // (a remainder of one or two units would run a whole sweep of the matrix kernel for 4-8 columns: split the tail evenly instead)
            if (nunits_all - u0 > umax && nunits_all - u0 < umax + (spin ? 3 : 5)) units = (nunits_all - u0 + 1) / 2;
            // a sweep that does not fit the free HBM (nside 8192: 213 + 129 GB for ten fields) is cut down; below the matrix kernel's
            // smallest useful batch the rest goes to the vector-unit kernel's sweeps of four maps / two fields
            while (units > 1 && sweep_bytes(units) > avail) --units;
            if (units < (spin ? 3 : 5) && sweep_bytes(units) > avail) units = 0;
            if (units < (spin ? 3 : 5) && (units == 0 || nunits_all - u0 >= (spin ? 3 : 5))) {
                HX_TRY(synthesis_batch_valu(pl, spin, (nunits_all - u0) * cpu, d_alms + (size_t)u0 * cpu * pl->nlm, d_maps + (size_t)u0 * cpu * pl->npix,
                                            d_ref ? d_ref + (size_t)u0 * cpu * pl->npix : nullptr));
                break;
            }
            const int nc = units * cpu, rowlen = synth_duo_rowlen(spin, units);
            const size_t fv_bytes = sizeof(double) * (size_t)(pl->lmax + 1) * pl->nrp_pad * rowlen;
            const size_t zc_bytes = sizeof(double2) * (size_t)pl->ny * nc;
            const size_t fv_pad = (fv_bytes + 255) & ~(size_t)255;
            HX_TRY(pl->F.alloc(fv_pad + zc_bytes));
            HX_TRY(pl->syn_tab.alloc(synth_duo_table_bytes(pl, spin, units)));
            double *fv = pl->F.as<double>();
            double2 *zc = reinterpret_cast<double2 *>(reinterpret_cast<char *>(pl->F.p) + fv_pad);
            HX_TRY(launch_synth_duo(pl, spin, units, *ts, d_alms + (size_t)u0 * cpu * pl->nlm, pl->syn_tab.as<double>(), fv));
            ProfScope ps("ring_fft");
            {
                // ring pairs whose rings hold every order in a bin of its own go through the transposing pass, the polar ones through the gather
                const int *ml = (spin ? pl->syn_mlim2 : pl->syn_mlim0).as<int>();
                int rp_t = pl->nrp;
                while (rp_t > 0 && 4 * pl->h_nsub[rp_t - 1] >= 2 * pl->lmax + 2) --rp_t;
                if (rp_t > 0)
                    hipLaunchKernelGGL(k_synth_spectrum_v, dim3((rp_t + 3) / 4), dim3(256), 0, st, P, fv, nc, pl->lmax, zc, ml, rp_t);
                if (rp_t < pl->nrp) {
                    const int nphi_max = 4 * pl->h_nsub[pl->nrp - 1];
                    const dim3 grid(pl->nrp - rp_t, (nphi_max / 2 + SPT_M) / SPT_M);
                    hipLaunchKernelGGL(k_synth_spectrum_t, grid, dim3(256), sizeof(double2) * 2 * nc * SPT_M, st, P, fv, nc, pl->lmax, zc, ml, rp_t);
                }
            }
            HX_TRY(launch_subdft_classes<1>(pl, nc, nullptr, nullptr, zc, nullptr, 0, 0x7fffffff, d_maps + (size_t)u0 * cpu * pl->npix,
                                            d_ref ? d_ref + (size_t)u0 * cpu * pl->npix : nullptr));
            u0 += units;
        }
        HX_HIP(hipGetLastError());
        return HX_OK;
    }
    return synthesis_batch_valu(pl, spin, nb, d_alms, d_maps, d_ref);
}

static int check_sht_args(hx_plan *pl, int spin, int ncomp, const void *a, const void *b)
{
    if (!pl || !a || !b) return fail(HX_ERR_ARG, "null plan or buffer");
    if (pl->nside < 1) return fail(HX_ERR_ARG, "not a HEALPix plan");
    if (spin != 0 && spin != 2) return fail(HX_ERR_UNSUPPORTED, "spin-%d maps not yet supported", spin);
    if (ncomp < 1 || (spin == 2 && (ncomp & 1))) return fail(HX_ERR_ARG, "bad component count %d for spin %d", ncomp, spin);
    return HX_OK;
}

namespace hx {
__global__ void k_apply_fl(int lmax, int ncomp, long long nlm, const double *__restrict__ fl, double2 *__restrict__ alm)
{
    const int m = blockIdx.x;
    for (int i = threadIdx.x; i < (lmax - m + 1) * ncomp; i += blockDim.x) {
        const int c = i / (lmax - m + 1), l = m + i % (lmax - m + 1);
        double2 *p = alm + c * nlm + almidx(lmax, l, m);
        p->x *= fl[l];
        p->y *= fl[l];
    }
}
}  // namespace hx

static int apply_fl(hx_plan *pl, int nb, double2 *alm, const double *fl)
{
    hipLaunchKernelGGL(k_apply_fl, dim3(pl->lmax + 1), dim3(256), 0, rt().stream, pl->lmax, nb, pl->nlm, fl, alm);
    HX_HIP(hipGetLastError());
    return HX_OK;
}

static int map2alm_multi_impl(hx_plan *pl, int njobs, const int *spins, const int *ncomps, const double *const *maps, const double *const *const *comp_maps,
                              double *const *alms, const double *ring_weights, const double *pix_weights, const double *const *fls);

extern "C" int hx_map2alm(hx_plan *pl, int spin, int ncomp, const double *maps, double *alms,
                          const double *ring_weights, const double *pix_weights, const double *fl, int niter)
{
    HX_TRY(ensure_ready());
    HX_TRY(check_sht_args(pl, spin, ncomp, maps, alms));
    if (niter < 0) return fail(HX_ERR_ARG, "niter < 0");
    InView vmaps, vrw, vpw, vfl;
    OutView valms;
    // Host maps without iterations go through the upload pipeline of hx_map2alm_multi (one job): sweep k + 1 is staged (pageable ->
    // pinned -> HBM, second stream, three plan-owned buffers) while the GPU transforms sweep k
    if (niter == 0 && !is_device_ptr(maps) && copy_stream() != nullptr)
        return map2alm_multi_impl(pl, 1, &spin, &ncomp, &maps, nullptr, &alms, ring_weights, pix_weights, &fl);
    HX_TRY(vmaps.bind(maps, sizeof(double) * (size_t)ncomp * pl->npix));
    HX_TRY(vrw.bind(ring_weights, sizeof(double) * pl->nrp));
    HX_TRY(vpw.bind(pix_weights, sizeof(double) * (size_t)pl->npix));
    HX_TRY(classify_pixel_weights(pl, vpw.as<double>()));
    HX_TRY(vfl.bind(fl, sizeof(double) * (pl->lmax + 1)));
    HX_TRY(valms.bind(alms, sizeof(double2) * (size_t)ncomp * pl->nlm));
    // residual maps of the Jacobi iterations: plan-owned scratch (no per-call hipMalloc)
    DevBuf &resid = pl->resid_maps;
    // the sweeps are sized by analysis_next_batch(); the synthesis of the Jacobi iterations takes the maps / fields of a sweep in its
    // own sweeps of four maps / two fields
    if (niter > 0) HX_TRY(resid.alloc(sizeof(double) * (size_t)analysis_max_batch(spin, ncomp) * pl->npix));
    for (int c0 = 0, nb = 0; c0 < ncomp; c0 += nb) {
        nb = analysis_next_batch(spin, ncomp - c0);
        const double *dm = vmaps.as<double>() + (size_t)c0 * pl->npix;
        double2 *da = valms.as<double2>() + (size_t)c0 * pl->nlm;
        // the filter fl is applied once, after the last iteration
        HX_TRY(analysis_batch(pl, spin, nb, dm, da, vrw.as<double>(), vpw.as<double>(), niter == 0 ? vfl.as<double>() : nullptr, 0));
        for (int it = 0; it < niter; ++it) {
            HX_TRY(synthesis_batch(pl, spin, nb, da, resid.as<double>(), dm));
            HX_TRY(analysis_batch(pl, spin, nb, resid.as<double>(), da, vrw.as<double>(), vpw.as<double>(), nullptr, 1));
        }
        if (niter > 0 && fl) HX_TRY(apply_fl(pl, nb, da, vfl.as<double>()));
    }
    HX_TRY(valms.finish());
    // staging buffers of host arguments are released on return: only an all-device call may stay asynchronous
    if (vmaps.tmp.p || vrw.tmp.p || vpw.tmp.p || vfl.tmp.p || valms.tmp.p) {
        HX_HIP(hipStreamSynchronize(rt().stream));
        return HX_OK;
    }
    return finish_call();
}

// Several transforms as ONE call (the loop of heracles/mapping.py:151-172 over the (field, bin) maps of a job): host maps of ALL
// jobs go through one upload pipeline (pageable -> pinned -> HBM, second stream) that the transforms follow, across job boundaries, so
// that the call costs its PCIe time plus very little.  A host job is cut into the sweeps that cost least per map (ten fields / ten maps
// at the bench size) and every such sweep runs as a StreamSweep (hx_sht_common.h): its rings are uploaded slab by slab -- for every
// component the block of northern rings of the slab and the block of their southern partners -- and the slab's ring FFTs, operand rows
// and completed ring groups are queued behind it; what is left behind the last byte is a twelfth of a sweep (21 ms at the bench size:
// 48 GB in 882 ms).  Sweeps that cannot be streamed (the small batches of the vector-unit kernels, HX_STREAM_SLABS=0) are uploaded whole,
// at most 5 spin-2 fields / 8 spin-0 maps at a time with the last one halved until it holds at most two units (round 3's pipeline: 960 ms).
// Callers put their large jobs first.  niter = 0 only (iterations need their maps resident).
// comp_maps (hx_map2alm_list): job j's components are SEPARATE arrays comp_maps[j][c] (maps[j] = the first of them); they are
// gathered into the staging buffer of their sweep -- host arrays through the pinned pipeline, device arrays by copies on the
// upload stream; neighbours in memory go as one transfer.
static int map2alm_multi_impl(hx_plan *pl, int njobs, const int *spins, const int *ncomps, const double *const *maps, const double *const *const *comp_maps,
                              double *const *alms, const double *ring_weights, const double *pix_weights, const double *const *fls)
{
    HX_TRY(ensure_ready());
    if (!pl || njobs < 1 || !spins || !ncomps || !maps || !alms) return fail(HX_ERR_ARG, "hx_map2alm_multi: bad arguments");
    for (int j = 0; j < njobs; ++j) HX_TRY(check_sht_args(pl, spins[j], ncomps[j], maps[j], alms[j]));
    // a job goes through the staging buffers if its maps are on the host, or scattered over separate arrays
    auto staged = [&](int j) { return !is_device_ptr(maps[j]) || (comp_maps && comp_maps[j]); };
    InView vrw, vpw;
    HX_TRY(vrw.bind(ring_weights, sizeof(double) * pl->nrp));
    HX_TRY(vpw.bind(pix_weights, sizeof(double) * (size_t)pl->npix));
    HX_TRY(classify_pixel_weights(pl, vpw.as<double>()));
    std::vector<InView> vfl(njobs);
    std::vector<OutView> valm(njobs);
    struct Sweep { int job, c0, nb; bool stream; };
    std::vector<Sweep> sweeps;
    // HX_STREAM_SLABS: slabs of rings per streamed sweep (default 12; 0 or 1: the sweeps of round 3 -- whole maps, 5 fields / 8 maps at most)
    static int want_slabs = -1;
    if (want_slabs < 0) { const char *e = getenv("HX_STREAM_SLABS"); want_slabs = e ? atoi(e) : 12; }
    bool any_host = false;
    for (int j = 0; j < njobs; ++j) {
        HX_TRY(vfl[j].bind(fls ? fls[j] : nullptr, sizeof(double) * (pl->lmax + 1)));
        HX_TRY(valm[j].bind(alms[j], sizeof(double2) * (size_t)ncomps[j] * pl->nlm));
        const bool host = staged(j);
        any_host = any_host || host;
        const int unit = spins[j] ? 2 : 1, cap = spins[j] ? 10 : 8;  // 5 spin-2 fields / 8 spin-0 maps: one full column group each
        for (int c0 = 0; c0 < ncomps[j];) {
            // host maps: the sweep that costs least per map, its rings uploaded and transformed slab by slab (StreamSweep) ...
            int nb = analysis_next_batch(spins[j], ncomps[j] - c0);
            const bool stream = host && want_slabs > 1 && copy_stream() != nullptr && analysis_can_stream(pl, spins[j], nb);
            if (host && !stream) {
                // ... or, where that is not possible (the small batches of the vector-unit kernels), whole maps in small sweeps
                nb = std::min(cap, ncomps[j] - c0);
                // spin 2: two even sweeps rather than a full and a small one (a sweep costs ~76 ms before its first column);
                // spin 0: a full group, then the rest -- small spin-0 sweeps run on the vector-unit kernel at 23 ms per map
                if (spins[j] && ncomps[j] - c0 > cap && ncomps[j] - c0 < 2 * cap) nb = ((ncomps[j] - c0) / unit + 1) / 2 * unit;
            }
            sweeps.push_back({j, c0, nb, stream});
            c0 += nb;
        }
    }
    if (any_host && staged(sweeps.back().job) && !sweeps.back().stream) {
        for (;;) {  // halve the last sweep until it holds at most two units
            Sweep &l = sweeps.back();
            const int unit = spins[l.job] ? 2 : 1, units = l.nb / unit;
            if (units <= 2) break;
            const int first = (units + 1) / 2 * unit;
            const Sweep tail = {l.job, l.c0 + first, l.nb - first, false};
            l.nb = first;
            sweeps.push_back(tail);
        }
    }
    hipStream_t cs = any_host ? copy_stream() : nullptr;
    size_t stage_bytes = 0;
    for (const Sweep &w : sweeps)
        if (staged(w.job)) stage_bytes = std::max(stage_bytes, (size_t)(sizeof(double) * (size_t)w.nb * (size_t)pl->npix));
    // host sweeps are numbered in upload order; buffer h % NST holds host sweep h.  THREE buffers: the upload of sweep h + 1 waits
    // for the transform of sweep h - 2, not h - 1 -- with two, a 4.8 GB upload sat 130 ms behind the 240 ms transform of the
    // spin-2 sweep before it (tools/time_host_multi.py).  Streamed sweeps are up to 32 GB each and end right behind their upload: two.
    const int NST = 3 * (double)stage_bytes > 64e9 ? 2 : hx_plan::NSTAGE;
    constexpr int NEV = hx_plan::NUNIT_EV;
    hipEvent_t *unit_up = pl->unit_up;
    if (any_host) {
        if (!cs) return fail(HX_ERR_HIP, "hx_map2alm_multi: no copy stream");
        for (int i = 0; i < NST; ++i) {
            HX_TRY(pl->stage[i].alloc(stage_bytes));
            if (!pl->stage_done[i]) HX_HIP(hipEventCreateWithFlags(&pl->stage_done[i], hipEventDisableTiming));
        }
        for (int i = 0; i < NEV; ++i)
            if (!unit_up[i]) HX_HIP(hipEventCreateWithFlags(&unit_up[i], hipEventDisableTiming));
    }
    std::vector<int> hidx(sweeps.size(), -1);
    int nh = 0;
    for (size_t k = 0; k < sweeps.size(); ++k)
        if (staged(sweeps[k].job)) hidx[k] = nh++;
    // streamed sweeps: slab edges, task tables and scratch of ALL of them before anything is queued
    std::vector<StreamSweep> ss(sweeps.size());
    for (size_t k = 0; k < sweeps.size(); ++k) {
        const Sweep &w = sweeps[k];
        if (!w.stream) continue;
        HX_TRY(analysis_stream_plan(pl, spins[w.job], w.nb, want_slabs, ss[k]));
        ss[k].d_maps = pl->stage[hidx[k] % NST].as<double>();
        ss[k].d_alms = valm[w.job].as<double2>() + (size_t)w.c0 * pl->nlm;
        ss[k].d_rw = vrw.as<double>(); ss[k].d_pw = vpw.as<double>(); ss[k].d_fl = vfl[w.job].as<double>();
    }
    // units of work in order: a whole sweep, or one slab of a streamed sweep; units of host sweeps have an upload in front of them
    struct Unit { size_t k; int slab; };
    std::vector<Unit> units;
    for (size_t k = 0; k < sweeps.size(); ++k) {
        if (sweeps[k].stream)
            for (int q = 0; q < ss[k].nslab; ++q) units.push_back({k, q});
        else
            units.push_back({k, -1});
    }
    std::vector<int> uidx(units.size(), -1);  // number of the unit among those with an upload
    int nu = 0;
    for (size_t u = 0; u < units.size(); ++u)
        if (hidx[units[u].k] >= 0) uidx[u] = nu++;
    // component c of a sweep: its source array and its place in the staging buffer
    auto comp_src = [&](const Sweep &w, int c) -> const double * {
        return (comp_maps && comp_maps[w.job]) ? comp_maps[w.job][w.c0 + c] : maps[w.job] + (size_t)(w.c0 + c) * pl->npix;
    };
    auto push = [&](void *dst, const double *src, size_t bytes) -> int {
        if (is_device_ptr(src)) HX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cs));
        else HX_TRY(copy_h2d(dst, src, bytes, cs));
        return HX_OK;
    };
    auto upload = [&](size_t u) -> int {
        const Sweep &w = sweeps[units[u].k];
        const int b = hidx[units[u].k] % NST, slab = units[u].slab;
        if (slab <= 0 && hidx[units[u].k] >= NST) HX_HIP(hipEventSynchronize(pl->stage_done[b]));  // host sweep h - NST has read this buffer
        double *stage = pl->stage[b].as<double>();
        if (slab < 0) {
            for (int c = 0; c < w.nb;) {  // runs of components that lie behind one another in memory: one transfer each
                int e = c + 1;
                while (e < w.nb && comp_src(w, e) == comp_src(w, e - 1) + pl->npix) ++e;
                HX_TRY(push(stage + (size_t)c * pl->npix, comp_src(w, c), sizeof(double) * (size_t)(e - c) * pl->npix));
                c = e;
            }
        } else {
            // the rings of the slab: a block of northern rings and the block of their southern partners, per component
            const StreamSweep &sw = ss[units[u].k];
            const int r0 = sw.rp_edge[slab], r1 = std::min(sw.rp_edge[slab + 1], pl->nrp) - 1;  // first and last ring pair
            const long long n0 = pl->h_startN[r0], n1 = pl->h_startN[r1] + 4LL * pl->h_nsub[r1];
            const int rs = pl->h_startS[r1] >= 0 ? r1 : r1 - 1;                                  // (the equator has no southern ring)
            const long long s0 = rs >= r0 ? pl->h_startS[rs] : 0, s1 = rs >= r0 ? pl->h_startS[r0] + 4LL * pl->h_nsub[r0] : 0;
            for (int c = 0; c < w.nb; ++c) {
                const double *src = comp_src(w, c);
                double *dst = stage + (size_t)c * pl->npix;
                HX_TRY(push(dst + n0, src + n0, sizeof(double) * (size_t)(n1 - n0)));
                if (s1 > s0) HX_TRY(push(dst + s0, src + s0, sizeof(double) * (size_t)(s1 - s0)));
            }
        }
        HX_HIP(hipEventRecord(unit_up[uidx[u] % NEV], cs));
        return HX_OK;
    };
    auto next_upload = [&](size_t u) -> size_t {  // first unit after u with an upload
        for (size_t q = u + 1; q < units.size(); ++q)
            if (uidx[q] >= 0) return q;
        return units.size();
    };
    // HX_TRACE=1: host-side timeline of the call on stderr (ms since entry): when each unit's upload was staged and issued
    const bool trace = getenv("HX_TRACE") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(); };
    auto traced_upload = [&](size_t u) -> int {
        const double t0 = now_ms();
        const int rc = upload(u);
        if (trace) {
            const Sweep &w = sweeps[units[u].k];
            fprintf(stderr, "[hx] multi: sweep %zu (job %d, spin %d, %d comps) slab %d staged %.0f -> %.0f ms\n", units[u].k, w.job, spins[w.job], w.nb, units[u].slab, t0, now_ms());
        }
        return rc;
    };
    const size_t first_up = next_upload((size_t)-1);
    if (first_up < units.size()) HX_TRY(traced_upload(first_up));
    for (size_t u = 0; u < units.size(); ++u) {
        const size_t k = units[u].k;
        const Sweep &w = sweeps[k];
        const int slab = units[u].slab;
        if (uidx[u] >= 0) HX_HIP(hipStreamWaitEvent(rt().stream, unit_up[uidx[u] % NEV], 0));
        bool done = true;
        if (slab < 0) {
            const double *src = hidx[k] >= 0 ? pl->stage[hidx[k] % NST].as<double>() : maps[w.job] + (size_t)w.c0 * pl->npix;
            HX_TRY(analysis_batch(pl, spins[w.job], w.nb, src, valm[w.job].as<double2>() + (size_t)w.c0 * pl->nlm, vrw.as<double>(), vpw.as<double>(),
                                  vfl[w.job].as<double>(), 0));
        } else {
            if (slab == 0) HX_TRY(analysis_stream_start(ss[k]));
            HX_TRY(analysis_stream_slab(ss[k], slab));
            done = slab + 1 == ss[k].nslab;
            if (done) HX_TRY(analysis_stream_end(ss[k]));
        }
        if (done && hidx[k] >= 0) HX_HIP(hipEventRecord(pl->stage_done[hidx[k] % NST], rt().stream));
        if (trace) fprintf(stderr, "[hx] multi: sweep %zu slab %d queued at %.0f ms\n", k, slab, now_ms());
        if (uidx[u] >= 0) {
            const size_t q = next_upload(u);
            if (q < units.size()) HX_TRY(traced_upload(q));  // the host thread stages the next unit while this one is transformed
        }
    }
    for (int j = 0; j < njobs; ++j) HX_TRY(valm[j].finish());
    if (trace) fprintf(stderr, "[hx] multi: everything queued at %.0f ms\n", now_ms());
    HX_HIP(hipStreamSynchronize(rt().stream));  // staging buffers of host arguments are released on return
    if (trace) fprintf(stderr, "[hx] multi: done at %.0f ms\n", now_ms());
    return HX_OK;
}

extern "C" int hx_map2alm_multi(hx_plan *pl, int njobs, const int *spins, const int *ncomps, const double *const *maps, double *const *alms,
                                const double *ring_weights, const double *pix_weights, const double *const *fls)
{
    return map2alm_multi_impl(pl, njobs, spins, ncomps, maps, nullptr, alms, ring_weights, pix_weights, fls);
}

// The loop of heracles/mapping.py:151-172 as ONE call over the arrays the reference holds: one array per map -- [npix] for spin 0,
// [2][npix] (Q, U) for spin 2 -- and one output array per map ([nlm] / [2][nlm] complex).  The maps are gathered sweep by sweep
// into the staging buffers of hx_map2alm_multi (no stacked copy on the host: np.stack of the bench's 48 GB costs several seconds),
// spin-2 fields first; the alms are collected in HBM and handed out at the end.  niter > 0 (Jacobi iterations need their maps
// resident): the maps of a spin are gathered into one device array first, then transformed as a batch.
extern "C" int hx_map2alm_list(hx_plan *pl, int nmaps, const int *spins, const double *const *maps, double *const *alms,
                               const double *ring_weights, const double *pix_weights, const double *fl0, const double *fl2, int niter)
{
    HX_TRY(ensure_ready());
    if (!pl || nmaps < 1 || !spins || !maps || !alms || niter < 0) return fail(HX_ERR_ARG, "hx_map2alm_list: bad arguments");
    std::vector<const double *> comps[2];  // [0]: spin 2, [1]: spin 0 (large jobs first)
    std::vector<int> owner[2];
    for (int i = 0; i < nmaps; ++i) {
        if (spins[i] != 0 && spins[i] != 2) return fail(HX_ERR_UNSUPPORTED, "spin-%d maps not yet supported", spins[i]);
        if (!maps[i] || !alms[i]) return fail(HX_ERR_ARG, "hx_map2alm_list: null map or alm %d", i);
        const int g = spins[i] ? 0 : 1;
        comps[g].push_back(maps[i]);
        if (spins[i]) comps[g].push_back(maps[i] + pl->npix);
        owner[g].push_back(i);
    }
    int jspin[2], jn[2], nj = 0;
    const double *jmaps[2], *jfl[2];
    const double *const *jcomp[2];
    double *jalm[2];
    DevBuf out[2];
    for (int g = 0; g < 2; ++g) {
        if (comps[g].empty()) continue;
        HX_TRY(out[g].alloc(sizeof(double2) * comps[g].size() * (size_t)pl->nlm));
        jspin[nj] = g == 0 ? 2 : 0; jn[nj] = (int)comps[g].size(); jmaps[nj] = comps[g][0]; jcomp[nj] = comps[g].data();
        jalm[nj] = out[g].as<double>(); jfl[nj] = g == 0 ? fl2 : fl0;
        ++nj;
    }
    if (niter == 0) {
        HX_TRY(map2alm_multi_impl(pl, nj, jspin, jn, jmaps, jcomp, jalm, ring_weights, pix_weights, jfl));
    } else {
        InView vrw, vpw;  // (bound once: a host weight array is not uploaded per spin)
        HX_TRY(vrw.bind(ring_weights, sizeof(double) * pl->nrp));
        HX_TRY(vpw.bind(pix_weights, sizeof(double) * (size_t)pl->npix));
        for (int j = 0; j < nj; ++j) {
            DevBuf in;
            HX_TRY(in.alloc(sizeof(double) * (size_t)jn[j] * pl->npix));
            for (int c = 0; c < jn[j];) {
                int e = c + 1;
                while (e < jn[j] && jcomp[j][e] == jcomp[j][e - 1] + pl->npix) ++e;
                char *dst = (char *)in.p + sizeof(double) * (size_t)c * pl->npix;
                const size_t bytes = sizeof(double) * (size_t)(e - c) * pl->npix;
                if (is_device_ptr(jcomp[j][c])) HX_HIP(hipMemcpyAsync(dst, jcomp[j][c], bytes, hipMemcpyDeviceToDevice, rt().stream));
                else HX_TRY(copy_h2d(dst, jcomp[j][c], bytes));
                c = e;
            }
            HX_TRY(hx_map2alm(pl, jspin[j], jn[j], in.as<double>(), jalm[j], vrw.as<double>(), vpw.as<double>(), jfl[j], niter));
            HX_HIP(hipStreamSynchronize(rt().stream));  // `in` is released here
        }
    }
    // (the call above has synchronised) alms out: host arrays through the pinned pipeline, device arrays by device copies
    for (int g = 0; g < 2; ++g) {
        const int cpu = g == 0 ? 2 : 1;
        for (size_t u = 0; u < owner[g].size(); ++u) {
            const size_t bytes = sizeof(double2) * (size_t)cpu * pl->nlm;
            const char *src = (const char *)out[g].p + u * bytes;
            double *dst = alms[owner[g][u]];
            if (is_device_ptr(dst)) HX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt().stream));
            else HX_TRY(copy_d2h(dst, src, bytes));
        }
    }
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

extern "C" int hx_alm2map(hx_plan *pl, int spin, int ncomp, const double *alms, double *maps)
{
    HX_TRY(ensure_ready());
    HX_TRY(check_sht_args(pl, spin, ncomp, alms, maps));
    InView valms;
    OutView vmaps;
    HX_TRY(valms.bind(alms, sizeof(double2) * (size_t)ncomp * pl->nlm));
    HX_TRY(vmaps.bind(maps, sizeof(double) * (size_t)ncomp * pl->npix));
    HX_TRY(synthesis_batch(pl, spin, ncomp, valms.as<double2>(), vmaps.as<double>(), nullptr));
    HX_TRY(vmaps.finish());
    if (valms.tmp.p || vmaps.tmp.p) {
        HX_HIP(hipStreamSynchronize(rt().stream));
        return HX_OK;
    }
    return finish_call();
}
