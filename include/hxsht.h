/* hxsht.h -- C ABI of libhxsht.so, the MI355X (gfx950) harmonic-space two-point engine.
 *
 * Drop-in boundary for the hot path of heracles-ec/heracles.  The
 * reference is pure Python and binds no native library itself; each entry point below
 * replaces the third-party / numpy call the reference makes at the cited file:line, and
 * INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every pointer argument may be a HOST pointer or a
 *    DEVICE (HBM) pointer; the library detects which (hipPointerGetAttributes) and stages
 *    host buffers through its own device scratch.  The caller owns all buffers.
 *  - Complex arrays are interleaved (re, im) doubles == numpy complex128.
 *  - alm layout: m-major, mmax == lmax, idx(l,m) = m*(2*lmax+1-m)/2 + l
 *    (heracles/twopoint.py:90-99, heracles/ducc.py:157-161).
 *  - Maps are HEALPix RING ordered, npix = 12*nside^2 (heracles/healpy.py:124-142).
 *  - Return value 0 = ok, negative = error; message via hx_last_error() (thread local).
 *  - Calls are synchronous on return unless hx_set_async(1) was called.
 *  - There is NO CPU fallback: every compute entry point fails with HX_ERR_NO_DEVICE when
 *    no gfx950 device is usable.
 *  - Threading: one host thread per process drives the library (one process per GPU).  Only hx_last_error() is
 *    thread local and only the host <-> device staging is serialised internally; plans, the runtime and the profile are not
 *    thread safe.  The library does not survive fork() (HIP does not): a child must not call into it.
 */
#ifndef HXSHT_H
#define HXSHT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HX_OK 0
#define HX_ERR_ARG (-1)
#define HX_ERR_NO_DEVICE (-2)
#define HX_ERR_HIP (-3)
#define HX_ERR_MEM (-4)
#define HX_ERR_UNSUPPORTED (-5)

typedef struct hx_plan hx_plan;

/* ---- runtime ------------------------------------------------------------------- */
const char *hx_version(void);
const char *hx_last_error(void);
int hx_device_count(void);
int hx_init(int device);          /* select device, create the library stream           */
int hx_set_stream(void *stream);  /* use a caller-owned hipStream_t (NULL = own stream)  */
void *hx_get_stream(void);
int hx_set_async(int on);         /* 1: do not synchronise before returning (device ptrs) */
int hx_synchronize(void);
/* dst <- src (bytes), either side host or device memory; host <-> device through the library's pinned staging pipeline (~55 GB/s
 * from / to pageable memory).  Complete on return. */
int hx_copy(void *dst, const void *src, int64_t bytes);
/* Page-locked host memory the CALLER owns: a destination (or source) of this kind is reached by DMA directly, without the staging
 * copy and without first-touch page faults -- the buffer a caller keeps and passes again as `out` of hx_mixmat_eb / hx_mixctx_apply
 * for every matrix of a loop (the reference hands each result to its `out` mapping and drops it, heracles/twopoint.py:393-397).
 * hx_host_free(NULL) is a no-op.  The memory stays valid until freed, across hx_init calls. */
int hx_host_alloc(int64_t bytes, void **out);
int hx_host_free(void *p);

/* HIP-event timers on the library stream (bench.py's timed region / roofline). */
int hx_timer_start(void);
int hx_timer_stop(float *ms);
/* per-kernel profile: accumulate HIP-event durations of every launch of the named kernel family:
 * "ring_fft", "fourier_combine", "legendre_analysis" (all analysis kernels; also split into "legendre_analysis_s0" /
 * "legendre_analysis_s2" by spin and "legendre_valu" for the single-map vector-unit kernel), "alm_reduce", "legendre_synthesis",
 * "alm2cl", "mixmat_gemm", "wigner_tables".  hx_profile_get returns launches and total milliseconds. */
int hx_profile_enable(int on);
int hx_profile_reset(void);
int hx_profile_get(const char *name, int *launches, double *total_ms);

/* ---- spherical harmonic transforms --------------------------------------------- */
/* Replaces healpy.map2alm / alm2map as called at heracles/healpy.py:183-189 and
 * heracles/io.py:377.  A plan owns ring tables, recursion tables, Bluestein tables and
 * device scratch for up to max_comp map components per call. */
hx_plan *hx_plan_create(int nside, int lmax, int max_comp);
/* Longest ring FFT kept in LDS by plans created afterwards (power of two in [16, 8192], default 8192); Bluestein rings of up to
 * twice that run as two half-length passes.  A tuning / test knob: results do not depend on it beyond rounding. */
int hx_set_max_lds_fft(int points);
void hx_plan_destroy(hx_plan *plan);
int64_t hx_plan_scratch_bytes(const hx_plan *plan);
/* Frees the plan's transient HBM scratch (ring spectra, Legendre operands and rows, staging buffers of host maps, residual maps,
 * synthesis tables: up to ~200 GB after a full-size job).  Everything is allocated again on demand; the tables of the plan stay. */
int hx_plan_release_scratch(hx_plan *plan);
/* HBM the analysis may use for the operands and ring-group partial sums of ONE m-chunk (bytes; a chunk always
 * holds at least one m).  0 (default) = min(80 GB, half of the free HBM); the environment variable
 * HX_SCRATCH_GB sets the initial value.  The result does not depend on the chunking (tests/test_gpu_sht.py). */
int hx_set_scratch_budget(double bytes);
/* m-chunks the most recent analysis sweep of this plan was cut into (diagnostic / tests). */
int hx_plan_last_chunks(const hx_plan *plan);
/* Measurement aid: what this device sustains, from micro-kernels repeated for ~0.25 s each (clock settled): out4 = { HBM read GB/s,
 * HBM copy GB/s (read + write), FP64 MFMA 16x16x4 TFLOP/s, FP64 VALU FMA TFLOP/s }.             */
int hx_measure_peaks(double *out4);
/* Shader clock in GHz held during the FP64 MFMA probe of the last hx_measure_peaks call (0 before any). */
double hx_measured_mfma_clock(void);
/* Shader clock in GHz held UNDER the mixing-matrix GEMM since the last call of this function (every 64th tile samples s_memtime /
 * s_memrealtime around its work); 0 when no such tile has run.  Measurement aid for bench.py's gemm_frac_of_fp64_mfma_peak. */
double hx_mixmat_gemm_clock(void);
/* Measurement aid (bench.py's roofline): matrix-instruction flops one hx_map2alm(niter = 0) of
 * ncomp components EXECUTES (task list x MFMAs per wave-block), as opposed to the algorithmic
 * 8 * 2 nside * nlm per component the roofline is quoted on.                              */
int hx_plan_mfma_flops(hx_plan *plan, int spin, int ncomp, double *flops);
/* out2[0] = the same, out2[1] = FP64 vector flops of the recursions (4 per value of lambda_lm(theta) generated). */
int hx_plan_executed_flops(hx_plan *plan, int spin, int ncomp, double *out2);
/* What the Legendre analysis kernels of this process EXECUTED since the last reset, counted by the kernels themselves (one
 * atomic per wave): out2[0] = FP64 flops of the matrix instructions issued (stages whose rings are all still dead -- below 2^-100 (spin 2) / 2^-300 (spin 0) -- issue
 * none -- the task-list figure above counts them), out2[1] = FP64 flops on the vector unit.  reset != 0 zeroes the counters.
 * bench.py's roofline.achieved is these / the kernels' HIP-event time; it agrees with the SQ_INSTS_VALU_MFMA_F64 counter. */
int hx_executed_flops(double *out2, int reset);

/* maps  : [ncomp][npix] double; spin 2: components come in (Q,U) pairs, ncomp even
 * alms  : [ncomp][nlm] complex; spin 2: (E,B) pairs
 * ring_weights : NULL or [2*nside] quadrature weights of ring pairs (north ring 1..2nside)
 * pix_weights  : NULL or [npix] per-pixel weights (healpy use_pixel_weights=True data).  An array that repeats over the four
 *                quadrants of every ring and from north to south -- healpy's weights do -- is recognised per call and read
 *                once per pixel pair instead of eight times; any other array is used as it is
 * fl    : NULL or [lmax+1] filter applied to the output, alm[l,m] *= fl[l]
 *         (pixel-window deconvolution, heracles/healpy.py:172-196)
 * niter : Jacobi refinement iterations (healpy's `iter`)                              */
int hx_map2alm(hx_plan *plan, int spin, int ncomp, const double *maps, double *alms,
               const double *ring_weights, const double *pix_weights, const double *fl,
               int niter);
int hx_alm2map(hx_plan *plan, int spin, int ncomp, const double *alms, double *maps);
/* The loop of heracles/mapping.py:151-172 (one transform per (field, bin) map) as ONE call over njobs transforms
 * (spins[j], ncomps[j], maps[j], alms[j], fls[j] as in hx_map2alm; fls may be NULL; niter = 0): host maps of all jobs share one
 * upload pipeline that the transforms follow slab of rings by slab of rings, so that the call costs its PCIe time plus a twelfth of one
 * sweep.  Put large jobs first.  Results are bit-identical to those of njobs hx_map2alm calls on the same (host or device) maps. */
int hx_map2alm_multi(hx_plan *plan, int njobs, const int *spins, const int *ncomps, const double *const *maps, double *const *alms,
                     const double *ring_weights, const double *pix_weights, const double *const *fls);
/* The same for SEPARATE arrays, as heracles holds them (heracles/mapping.py:151-172: one array per (field, bin)): map i is
 * maps[i] -- [npix] for spin 0, [2][npix] (Q, U) for spin 2 --, its alm goes to alms[i] ([nlm] / [2][nlm] complex); host or
 * device pointers.  The maps are gathered sweep by sweep into the upload pipeline of hx_map2alm_multi (no stacked host copy),
 * spin-2 fields first; fl0 / fl2: NULL or the [lmax+1] filter of the spin-0 / spin-2 maps.  niter > 0: the maps of a spin are
 * gathered into one device array first (iterations need them resident) and transformed as a batch. */
int hx_map2alm_list(hx_plan *plan, int nmaps, const int *spins, const double *const *maps, double *const *alms,
                    const double *ring_weights, const double *pix_weights, const double *fl0, const double *fl2, int niter);

/* ---- the two halves of hx_map2alm, for the m-sharded multi-GPU route (heracles/mapping.py:151-172 and
 * heracles/twopoint.py:198-215 are the loops it shards) -------------------------------------------------------------------
 * A set of orders is (first, count, step): m = first + k step, k < count; rank q of N owns (q, ., N).
 * hx_ring_modes: ring Fourier stage of ncomp maps; for every set q < nsets = (m_first[q], m_count[q], m_step) the block
 *   outs[q][comp][k][nrp_pad][4] = (F_N.re, F_N.im, F_S.re, F_S.im)(m, ring pair) with ring phase, pixel and ring weights
 *   applied (DEVICE buffers of ncomp * hx_ring_modes_size(count) doubles) -- what is sent to the rank that owns the set.
 * hx_legendre_from_modes: Legendre stage of ALL ncomp components of one spin for one set of orders: comp_modes[c] points to
 *   component c's block [count][nrp_pad][4] (as received); writes alm[c][idx(l, m)] for the orders of the set ONLY (device buffer). */
int64_t hx_ring_modes_size(const hx_plan *plan, int count);
int hx_ring_modes(hx_plan *plan, int ncomp, const double *maps, const double *pix_weights, const double *ring_weights, int nsets,
                  const int *m_first, const int *m_count, int m_step, double *const *outs);
int hx_legendre_from_modes(hx_plan *plan, int spin, int ncomp, const double *const *comp_modes, int m_first, int m_count, int m_step,
                           double *alms, const double *fl);

/* All-gather of alm shards over RCCL for hosts WITHOUT torch.distributed ("one ncclAllGather of the alm shards
 * so every rank holds all alms"; the loops being sharded are heracles/mapping.py:151-172 and heracles/twopoint.py:198-215).  In place on
 * the buffer hx_map2alm wrote: buf (device, interleaved complex) holds sum(counts) elements, rank q's shard of counts[q] complex
 * elements at offset sum_{p<q} counts[p], this rank's own shard already there; shards may differ in length (one ncclBroadcast per shard
 * in a group: over xGMI each shard travels on all links of its owner at once).  comm: an ncclComm_t the HOST created (ncclCommInitRank;
 * the library does no bootstrap); RCCL is bound at the first call (librccl.so.1), HX_ERR_UNSUPPORTED if it cannot be loaded.  The Python
 * layer does not use this: its collectives are torch.distributed's (heracles_amd/distributed.py). */
int hx_allgather_alms(void *comm, int nranks, const int64_t *counts, double *buf);

/* ---- two-point reduction -------------------------------------------------------- */
/* Replaces heracles.twopoint.alm2cl (heracles/twopoint.py:63-101) for a whole list of
 * component pairs in one launch.  alms[i] points to component i with lmax_i[i];
 * cls is [npairs][lmax_out+1]; requires lmax_out <= min over used components.         */
int hx_alm2cl_pairs(int ncomp, const int *lmax_i, const double *const *alms, int lmax_out,
                    int npairs, const int *pair_i, const int *pair_j, double *cls);
/* The same sum over the orders m0, m0 + mstep, ... < m1 only (still divided by 2l + 1): partial spectra of disjoint sets add up to
 * hx_alm2cl_pairs -- a rank's contribution on the m-sharded multi-GPU route (the loop of heracles/twopoint.py:90-99 cut by m). */
int hx_alm2cl_pairs_range(int ncomp, const int *lmax_i, const double *const *alms, int lmax_out, int npairs, const int *pair_i,
                          const int *pair_j, int m0, int m1, int mstep, double *cls);

/* ---- mixing matrices / Wigner-d ------------------------------------------------- */
/* Gauss-Legendre nodes (ascending) and weights; the `gauss_legendre` hook of
 * heracles/transforms.py:25-43. */
int hx_gauss_legendre(int n, double *x, double *w);
/* The same with every node as a double-double: node k = x[k] + xlo[k], |xlo[k]| <~ 1e-16.  The tables behind hx_mixmat / hx_mixmat_eb
 * (convolvecl.mixmat / mixmat_eb at heracles/twopoint.py:378-388) are evaluated at x + xlo: next to the poles d^l(x)' ~ l^2 / 2, and a
 * node rounded to a double shifts a matrix element at l ~ 4000 by 1e-11 of the largest
 * (tests/test_gpu_mixmat.py::test_mixmat_blocks_at_high_l_vs_3j). */
int hx_gauss_legendre_dd(int n, double *x, double *w, double *xlo);

/* D[k][l] = d^l_{ab}(x_k), l = 0..lmax, (a,b) in {(0,0),(2,0),(2,2),(2,-2),(1,1),(-1,1)} (zero below max(|a|,|b|));
 * out is [n][lmax+1] row-major.  (Functions of heracles/transforms.py:46-112: P_l, d20, d22, d2m2, d11, dm11.)       */
int hx_wigner_d_table(int lmax, int a, int b, int n, const double *x, double *out);

/* Replaces convolvecl.mixmat / mixmat_eb as called at heracles/twopoint.py:378-388.
 * cl has ncl entries (mask spectrum, zero-extended to l3max).  out: [l1max+1][l2max+1]
 * (mixmat) or [3][l1max+1][l2max+1] (mixmat_eb: EE->EE, EE->BB, EB->EB).              */
int hx_mixmat(const double *cl, int ncl, int l1max, int l2max, int l3max, int s1, int s2,
              double *out);
int hx_mixmat_eb(const double *cl, int ncl, int l1max, int l2max, int l3max, double *out);
/* hx_mixmat / hx_mixmat_eb / hx_mixmat_batch keep the mask-independent part of their last build (nodes and Wigner-d tables of one
 * (l1max, l2max, l3max)) and the staging buffer of a host destination in HBM between calls (~3 GB at L = 6144), so that the next
 * build of that size allocates nothing; this frees them.  (The reference rebuilds everything per convolvecl call,
 * heracles/twopoint.py:378-388.) */
int hx_mixmat_release(void);
/* Everything the library keeps in HBM between calls OUTSIDE a plan or a context -- none of it is counted by hx_plan_scratch_bytes: the
 * cache of hx_mixmat / hx_mixmat_eb / hx_mixmat_batch (above; its staging buffer of a host destination also serves hx_mixctx_apply and
 * outlives hx_mixctx_destroy) and the tables, partial sums and staging buffer hx_alm2cl_pairs keeps (<= 512 MB).  A long-lived host
 * process of the reference's loops (heracles/twopoint.py:173-299, :316-401) calls this between stages to hand the HBM back. */
int hx_release_caches(void);
/* ---- FITS wire format of maps and alms (heracles/io.py:128-218, "next" row) ------------------------------------
 * Payload conversion between a FITS binary table of 'D' columns (row-major, big-endian) and the component-major
 * native arrays of the path, one pass on the GPU; `table` / `array` host or device.
 *   table element (row, c1, c2) <-> array[c1*s1 + c2*s2 + row*srow]
 *   maps  (io.py:128-168, ncols columns):      nc1 = ncols, nc2 = 1, s1 = nrows, srow = 1
 *   alms  (io.py:189-218, columns real, imag): nc1 = 2, nc2 = r, s1 = 1, s2 = 2*nrows, srow = 2  (complex128 (r, nrows)) */
int hx_fits_unpack_f64(int64_t nrows, int nc1, int nc2, int64_t s1, int64_t s2, int64_t srow, const void *table, double *array);
int hx_fits_pack_f64(int64_t nrows, int nc1, int nc2, int64_t s1, int64_t s2, int64_t srow, const double *array, void *table);

/* ---- jackknife loop in HBM (heracles/dices/jackknife.py:93-248, "next" row) ------------------------------------
 * Region maps (jackknife.py:253-263: every pixel outside region k set to zero) and delete-k alms (jackknife.py:222-233,
 * :298-304: full alms minus the sum of the deleted regions' alms) without leaving the device.                      */
int hx_region_maps(int64_t npix, int ncomp, const double *maps, const double *region, double k, double *out);
int hx_alm_subtract(int64_t n, const double *full, int nsub, const double *const *subs, double *out);

/* The loop of heracles/twopoint.py:354-397 (one convolvecl call per mask pair and spin combination) as an object:
 * Gauss-Legendre nodes, Wigner-d tables and the GEMM tile list are built ONCE per (l1max, l2max, l3max); every mask streamed
 * through then costs its node weights and one GEMM per product (for jobs whose matrices do not fit in memory together).
 * kind 1: spin (0,0), 2: spin (0,2)/(2,0) -> out (l1max+1, l2max+1); 4: spin (2,2) -> out (3, l1max+1, l2max+1). */
/* The products of heracles.twopoint.apply_mixing_matrix (heracles/twopoint.py:497-524, `_M @ cl`): y[v] = M x[v] for nvec spectra
 * x [nvec][m] -> y [nvec][n], M row-major (n, m); every pointer host or device (a device-resident matrix is read once per four
 * spectra: HBM-bound).  Rows are summed in a fixed order (bitwise repeatable).                                        */
int hx_matvec(int n, int m, const double *M, int nvec, const double *x, double *y);

/* np.linalg.pinv(M, rcond) as heracles.twopoint.invert_mixing_matrix calls it (heracles/twopoint.py:447-460): out (m x n) = pseudo-inverse
 * of M (n x m, row-major), singular values <= rcond * the largest are dropped.  Blocked one-sided Jacobi SVD on the GPU (Gram matrices of
 * column-block pairs, their 64 x 64 eigenproblems in LDS, rotations), the final product on the FP64 matrix unit.  M / out host or device;
 * info (nullable, 4 doubles on the host): sweeps, singular values kept, largest, smallest kept.  Fails (HX_ERR_UNSUPPORTED, nothing
 * written to out) if the sweeps run out before the columns are orthogonal to rounding.  Unlike hx_matvec / hx_alm2cl_pairs the result
 * is NOT bitwise repeatable run to run: the Gram sums are unordered f64 atomics (differences at the 1e-16 level of the sums).       */
int hx_pinv(int n, int m, const double *M, double rcond, double *out, double *info);

typedef struct hx_mixctx hx_mixctx;
hx_mixctx *hx_mixctx_create(int l1max, int l2max, int l3max);
int hx_mixctx_apply(hx_mixctx *ctx, const double *cl, int ncl, int kind, double *out);
void hx_mixctx_destroy(hx_mixctx *ctx);
/* The rows of those matrices BINNED, as heracles.twopoint.mixing_matrices(..., bins, weights) returns them (heracles/twopoint.py:391-397:
 * heracles.result.binned along axis -2, heracles/result.py:124-248 -- what heracles/cli.py:696-716 asks for whenever the configuration
 * has `bins`, e.g. `bins = 32 log 2l+1` in examples/heracles.cfg:4-6).  Binning is linear in the rows, so the numerators are formed as
 * (binned Wigner-d tables) x diag(w xi) x (tables)^T -- an (nbins x N)(N x (l2max+1)) product instead of the full matrix followed by a
 * host loop over its columns (5 s per (3, 6145, 6145) matrix in the reference's function), and only nbins x (l2max+1) doubles leave the GPU.
 *   hx_mixctx_set_bins: which [l1max+1] = bin of output multipole l (0 .. nbins-1; -1: in no bin), w [l1max+1] = its weight,
 *     norm [nbins] = the divisor of every bin (the reference: the summed weight); host arrays; the binned tables follow lazily.
 *   hx_mixctx_apply_binned: kind as hx_mixctx_apply; out (nbins, l2max+1) or (3, nbins, l2max+1), host or device;
 *     out[b][l2] = sum_{l in b} w_l M[l][l2] / norm[b], exactly 0 where the sum is exactly 0 (heracles/result.py:132-135). */
int hx_mixctx_set_bins(hx_mixctx *ctx, int nbins, const int *which, const double *w, const double *norm);
int hx_mixctx_apply_binned(hx_mixctx *ctx, const double *cl, int ncl, int kind, double *out);
/* The same loop as ONE call over a list of masks:
 *   cls   [nmask][ncl] mask spectra; kinds [nmask] bit mask: 1 -> spin (0,0) into out00[k]; 2 -> spin (0,2)/(2,0) into
 *   out02[k]; 4 -> spin (2,2) into outeb[k] (3 matrices as hx_mixmat_eb).  Output pointers host or device. */
int hx_mixmat_batch(int nmask, const double *cls, int ncl, int l1max, int l2max, int l3max, const int *kinds,
                    double *const *out00, double *const *out02, double *const *outeb);

/* Replaces heracles.transforms._cl2corr / _corr2cl (heracles/transforms.py:115-204) for
 * nspec spectra at once: cls [nspec][lmax+1][4] <-> corrs [nspec][lmax+1][4].          */
int hx_cl2corr(int lmax, int nspec, const double *cls, double *corrs);
int hx_corr2cl(int lmax, int nspec, const double *corrs, double *cls);

/* Replaces the m-block loop of DiscreteMapper.resample (heracles/ducc.py:145-162): re-packs m-major
 * alms from band limit lmax_in to lmax_out (truncate or zero-pad).  alm_in: [ncomp][nlm(lmax_in)]
 * complex, alm_out: [ncomp][nlm(lmax_out)] complex.                                             */
int hx_alm_resample(int lmax_in, int lmax_out, int ncomp, const double *alm_in, double *alm_out);

/* ---- values at arbitrary points -> alm -----------------------------------------------------------
 * Replaces ducc0.sht.adjoint_synthesis_general(map=values, spin=spin, lmax=lmax, loc=loc, epsilon=epsilon) as called by
 * DiscreteMapper.map_values (heracles/ducc.py:121-128):  alm[c][idx(l,m)] = sum_p map[c][p] conj(sY_lm(loc[p])), m >= 0,
 * spin 0: every row of `map` on its own; spin 2: rows (Q, U) -> (E, B) with healpy's sign conventions (those of hx_map2alm).
 * loc: (npoints, 2) colatitude, longitude in radians (0 <= colatitude <= pi, else HX_ERR_ARG); map: (ncomp, npoints);
 * alm: (ncomp, nlm) complex, OVERWRITTEN (the caller adds it to its running sum, ducc.py:133); pointers host or device.
 * The object holds the non-uniform FFT of accuracy epsilon (heracles: 1e-12 for float64 values, 1e-5 for float32,
 * ducc.py:107-114) and the Legendre plan on equidistant rings; lmax <= 8191.  info4 = {lmax, N, n1, W}: rings on the
 * full circle, oversampled grid points per dimension, kernel width in cells.                                           */
typedef struct hx_pointsht hx_pointsht;
hx_pointsht *hx_pointsht_create(int lmax, double epsilon);
void hx_pointsht_destroy(hx_pointsht *ps);
int hx_pointsht_info(const hx_pointsht *ps, int *info4);
int hx_pointsht_adjoint(hx_pointsht *ps, int spin, int ncomp, int64_t npoints, const double *loc, const double *map, double *alm);

/* ---- catalogue -> map accumulation (first "next" row of SURVEY.md 8f) ---------------
 * hx_ang2pix_ring replaces hp.ang2pix(nside, lon, lat, lonlat=True) at
 * heracles/healpy.py:157 (RING scheme, lon/lat in degrees, n points).
 * hx_map_values replaces HealpixMapper.map_values (heracles/healpy.py:144-160) with the
 * compiled loop _map (heracles/healpy.py:58-66): maps[v][ipix[j]] += values[v][j] for
 * j = 0..n-1 IN THAT ORDER per pixel (bit-identical sums; flags = 0).  values: [nval][n],
 * maps: [nval][12 nside^2], read-modify-write, host or device.  HX_MAP_ATOMIC gives up the
 * ordering guarantee for one pass of hardware f64 atomics.                             */
#define HX_MAP_ATOMIC 1
int hx_ang2pix_ring(int nside, int64_t n, const double *lon, const double *lat, int64_t *ipix);
int hx_map_values(int nside, int64_t n, const double *lon, const double *lat, int nval,
                  const double *values, double *maps, int flags);

/* Replaces hp.ud_grade(data, nside, dtype=float64) of HealpixMapper.resample
 * (heracles/healpy.py:205-209): RING in, RING out, pess=False, power=None -- degrade = mean of
 * the unmasked children (UNSEEN if none), summed in numpy's pairwise order over the NEST
 * children; upgrade = replication.  in: [nmaps][12 nside_in^2], out: [nmaps][12 nside_out^2].
 * Both nside must be powers of two (healpy raises ValueError otherwise; here HX_ERR_ARG).   */
int hx_ud_grade(int nside_in, int nside_out, int nmaps, const double *in, double *out);

/* NESTED <-> RING reordering of full-sky maps: what hp.read_map does to a NESTED file before heracles.io.read_vmap sees it
 * (heracles/io.py:360-365).  to_ring != 0: in is NESTED, out RING; 0: the reverse.  in / out [nmaps][12 nside^2], host or device,
 * not in place.  nside a power of two.                                                                              */
int hx_reorder(int nside, int to_ring, int nmaps, const double *in, double *out);

/* healpy's pixel-weight files (`healpix_full_weights_nside_NNNN.fits`, the data hp.map2alm(use_pixel_weights=True, datapath=...)
 * of heracles/healpy.py:183-189 reads): expansion of the compressed half-quadrant weights -- hx_pixel_weights_size(nside) =
 * (nside + 1)(3 nside + 1) / 4 values -- to the full-sky array [12 nside^2] of multiplicative pixel weights 1 + w that hx_map2alm
 * takes as `pix_weights`.  compressed / weights host or device.                                                           */
int64_t hx_pixel_weights_size(int nside);
int hx_pixel_weights_expand(int nside, int64_t ncompressed, const double *compressed, double *weights);

#ifdef __cplusplus
}
#endif
#endif
