// Device kernels of the stage that consumes the likelihood matrix: genotype
// log-likelihoods, their normalisation, the EM for haplotype frequencies with
// the per-individual genotype calls, and variant posteriors
// (platypus/cpopulation.pyx:658-890 and :515-632).  All fp64.
//
// A window w has n_hap haplotypes, n_ind individuals and a dense likelihood
// matrix like[like_base + h * n_reads + r]; individual i owns the reads
// [ioff[i], ioff[i+1]).  Genotypes are the pairs (a <= b) in the order of
// platypus/cgenotype.pyx:90-96; packed index of (a, b): a n - a (a - 1) / 2 + b - a.
//
//   k_gl          log-likelihood of every (individual, genotype), maximum per window
//   k_gl_tile     the same for windows with many haplotypes, likelihood columns staged in LDS
//   k_norm        exp(ll - maximum of the window); also written as full symmetric
//                 n x n matrices, the layout the EM streams
//   k_em          EM iterations and genotype calls, one workgroup per window
//   k_em_small    the same for windows of at most 8 haplotypes, one wavefront per window
//   k_em_init/step/update/calls  the same for few large windows, one launch pair per iteration
//   k_var_post    posterior of a variant given the haplotypes that carry it
//
// Sums of logarithms over reads are taken as the logarithm of a product (see
// LogProduct); sums over genotypes are regrouped (see k_em).  Results agree
// with the reference to rounding, not bit for bit; every reduction has a fixed
// shape, so runs are reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

struct WinDesc {
    int32_t n_hap, n_ind, n_reads, n_gen;
    int32_t ld, reserved;  // row length of the full matrices: n_hap, or n_hap rounded up to whole 128-byte lines when
                           // the matrix is wider than a wavefront (rows then start on line boundaries)
    int64_t like_base;   // pair index of (haplotype 0, read 0)
    int64_t ioff_base;   // into the concatenated per-individual read offsets (n_ind + 1 entries)
    int64_t cell_base;   // into per-(individual, genotype) arrays: sum of n_ind * n_gen of the windows before
    int64_t full_base;   // into the full matrices: sum of n_ind * n_hap * ld before
    int64_t hap_base;    // into per-haplotype outputs
    int64_t ind_base;    // into per-individual outputs
};

// Windows that k_em_exact takes (EM sums in the reference's own order): up to 8 haplotypes whatever the number of
// individuals (the host checks that the largest of them fits the LDS), and up to 16 haplotypes while the window's
// values fit 16 KiB (a single sample's windows always do).  Host and kernels share this predicate.
constexpr int kExactMaxHaps = 8, kExactMaxHapsLight = 16, kExactLightCells = 2000;
__host__ __device__ inline bool em_is_exact(int n_hap, int n_ind, int n_gen) {
    return n_hap <= kExactMaxHaps || (n_hap <= kExactMaxHapsLight && (long long)n_ind * (n_gen + 1) <= kExactLightCells);
}
constexpr int kExactPad = 96;      // doubles of a wavefront's LDS slice beside the values: frequencies [16], genotype table [136 ints]

__device__ __forceinline__ int64_t tri_row_start(int a, int n) { return (int64_t)a * n - (int64_t)a * (a - 1) / 2; }

// row a of packed index g: largest a with tri_row_start(a) <= g
__device__ __forceinline__ int tri_row_of(int g, int n) {
    const double t = 2.0 * n + 1.0;
    int a = (int)((t - sqrt(t * t - 8.0 * (double)g)) * 0.5);
    a = a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
    while (a > 0 && tri_row_start(a, n) > g) --a;
    while (a + 1 < n && tri_row_start(a + 1, n) <= g) ++a;
    return a;
}

// Cross-lane reductions over a wavefront, result in every lane.  The partner of a lane is
// fetched with DPP inside its row of 16 (quad swaps, then mirrors: once the lanes of a group
// agree, any lane of the partner group will do) and the four row results are combined
// through lane reads -- a fixed balanced tree, an order of magnitude less latency than
// six LDS-crossbar shuffles.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);       // row_half_mirror
    v += dpp_f64<0x140>(v);       // row_mirror
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    return fmax(fmax(lane_f64(v, 0), lane_f64(v, 16)), fmax(lane_f64(v, 32), lane_f64(v, 48)));
}

// Order-preserving map of a double onto an unsigned 64-bit integer, so that the
// maximum of doubles can be taken with an integer atomic (any order, same result).
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_bits(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ int window_of_cell(const WinDesc* __restrict__ wins, int n_win, int64_t c) {
    int lo = 0, hi = n_win - 1;                      // last window with cell_base <= c
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (wins[mid].cell_base <= c) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// sum over reads of log(0.5 (l1[r] + l2[r])) as ONE logarithm: the terms' mantissas are multiplied
// (each factor in [0.5, 1), renormalised every 128 factors), their binary exponents added as
// integers, and log(m) + e ln 2 taken at the end.  A dozen instructions per read instead of a
// double-precision log(); the result differs from the reference's sum of logarithms
// (platypus/cgenotype.pyx:58-61) by a few units in the last place of either.
struct LogProduct {
    double m = 1.0;
    int e = 0, k = 0;
    __device__ __forceinline__ void mul(double l1, double l2) {
        const double x = l1 + l2;                       // the factor 0.5 is one off the exponent
        m *= __builtin_amdgcn_frexp_mant(x);
        e += __builtin_amdgcn_frexp_exp(x) - 1;
        if ((++k & 127) == 0) { e += __builtin_amdgcn_frexp_exp(m); m = __builtin_amdgcn_frexp_mant(m); }
    }
    __device__ __forceinline__ double log_value() const { return log(m) + (double)e * 0.6931471805599453094; }
};

__global__ __launch_bounds__(256) void k_fill_u64(unsigned long long* __restrict__ dst, int n, double v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = ordered_bits(v);
}

// ---------------------------------------------------------------------------
// k_gl: ll[cell] = sum over the individual's reads of
// log(0.5 (L[a][r] + L[b][r]))  (platypus/cgenotype.pyx:58-61), as one logarithm per cell.  One thread per
// (window, individual, genotype) cell; an individual without reads keeps 0.0
// and stays out of the window's maximum (platypus/cpopulation.pyx:717-723).
// win_max[w] must hold ordered_bits(-1e7) on entry (platypus/cpopulation.pyx:696).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gl(const WinDesc* __restrict__ wins, int n_win,
                                            const int32_t* __restrict__ ioff,
                                            const double* __restrict__ like,
                                            double* __restrict__ ll, unsigned long long* __restrict__ win_max,
                                            int64_t n_cells)
{
    // threads past the end redo the last cell (no stores), so that every lane of a wavefront
    // takes part in the cross-lane maximum below
    const int64_t c_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = c_raw < n_cells;
    const int64_t c = valid ? c_raw : n_cells - 1;
    const int w = window_of_cell(wins, n_win, c);
    const WinDesc W = wins[w];
    const int64_t rem = c - W.cell_base;
    const int i = (int)(rem / W.n_gen), g = (int)(rem % W.n_gen);
    const int a = tri_row_of(g, W.n_hap), b = a + (int)(g - tri_row_start(a, W.n_hap));
    const int r0 = ioff[W.ioff_base + i], nr = ioff[W.ioff_base + i + 1] - r0;
    const double* l1 = like + W.like_base + (int64_t)a * W.n_reads + r0;
    const double* l2 = like + W.like_base + (int64_t)b * W.n_reads + r0;
    LogProduct lp;
    for (int r = 0; r < nr; ++r) lp.mul(l1[r], l2[r]);
    const double s = lp.log_value();
    if (valid) ll[c] = s;
    // one atomic per wavefront when all its cells belong to one window, else one per cell
    const int w0 = __builtin_amdgcn_readfirstlane(w);
    if (__all(w == w0)) {
        const double m = wave_max(nr > 0 ? s : -1e7);
        if ((threadIdx.x & 63) == 0) atomicMax(&win_max[w0], ordered_bits(m));
    } else if (nr > 0) {
        atomicMax(&win_max[w], ordered_bits(s));
    }
}

// ---------------------------------------------------------------------------
// k_gl_tile: the same sums for windows with many haplotypes.  One workgroup per
// (window, individual): the individual's likelihood columns L[h][r] are staged
// in LDS in chunks of reads (row stride odd, so the lanes of a wavefront, which
// hold consecutive second haplotypes b, fall into different banks), and every
// thread walks the genotypes g = tid, tid + 256, ... keeping (a, b) by
// increments.  Reads are taken in read order, chunk after chunk (one logarithm
// per chunk; partial sums rest in ll between chunks).  k_gl reads L straight from global memory, with
// every lane of a wavefront on a different cache line; here the reads of L are
// coalesced once and the 2 x n_gen x n_reads operand fetches hit LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gl_tile(const WinDesc* __restrict__ wins, int n_win,
                                                 const int32_t* __restrict__ ioff,
                                                 const double* __restrict__ like,
                                                 double* __restrict__ ll, unsigned long long* __restrict__ win_max,
                                                 int lds_doubles)
{
    extern __shared__ double gl_lds[];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    int lo = 0, hi = n_win - 1;                      // last window with ind_base <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (wins[mid].ind_base <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int w = lo;
    const WinDesc W = wins[w];
    const int i = (int)(blockIdx.x - W.ind_base), n = W.n_hap;
    const int r0 = ioff[W.ioff_base + i], nr = ioff[W.ioff_base + i + 1] - r0;
    double* out = ll + W.cell_base + (int64_t)i * W.n_gen;
    if (nr == 0) {                                   // cpopulation.pyx:717-718
        for (int g = tid; g < W.n_gen; g += 256) out[g] = 0.0;
        return;
    }
    int chunk = lds_doubles / n;                     // reads per chunk
    chunk = chunk > nr ? nr : chunk;
    if (!(chunk & 1) && chunk > 1) --chunk;          // odd row stride
    const int stride = chunk | 1;
    const double* L = like + W.like_base + r0;
    double mx = -1e7;
    for (int rc = 0; rc < nr; rc += chunk) {
        const int cn = nr - rc < chunk ? nr - rc : chunk;
        __syncthreads();
        for (int k = tid; k < n * cn; k += 256) {
            const int h = k / cn, r = k - h * cn;
            gl_lds[h * stride + r] = L[(int64_t)h * W.n_reads + rc + r];
        }
        __syncthreads();
        int a = 0, p = tid;                          // genotype (a, a + p)
        for (int g = tid; g < W.n_gen; g += 256) {
            while (p >= n - a) { p -= n - a; ++a; }
            const double* l1 = gl_lds + a * stride;
            const double* l2 = gl_lds + (a + p) * stride;
            LogProduct lp;
            for (int r = 0; r < cn; ++r) lp.mul(l1[r], l2[r]);                  // cgenotype.pyx:58-61
            const double s = (rc == 0 ? 0.0 : out[g]) + lp.log_value();
            out[g] = s;
            if (rc + cn >= nr) mx = fmax(mx, s);
            p += 256;
        }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) atomicMax(&win_max[w], ordered_bits(fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]))));
}

// ---------------------------------------------------------------------------
// k_norm: platypus/cpopulation.pyx:732-734.  post = exp(ll - max of the window)
// for every cell, in place, and mirrored into the full matrices
// full[i][a][b] = full[i][b][a].  One thread per cell.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_norm(const WinDesc* __restrict__ wins, int n_win,
                                              const unsigned long long* __restrict__ win_max,
                                              double* __restrict__ cells, double* __restrict__ full,
                                              int64_t n_cells)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    const int w = window_of_cell(wins, n_win, c);
    const WinDesc W = wins[w];
    const int n = W.n_hap;
    const int64_t rem = c - W.cell_base;
    const int i = (int)(rem / W.n_gen), g = (int)(rem % W.n_gen);
    const double p = exp(cells[c] - from_ordered_bits(win_max[w]));
    cells[c] = p;
    const int a = tri_row_of(g, n), b = a + (int)(g - tri_row_start(a, n));
    double* F = full + W.full_base + (int64_t)i * n * W.ld;
    F[(int64_t)a * W.ld + b] = p;
    F[(int64_t)b * W.ld + a] = p;
}

// ---------------------------------------------------------------------------
// k_norm_tile: the same for windows with many haplotypes.  A workgroup takes a
// 32 x 32 tile (a, b) of the upper triangle of one individual's matrix: rows of
// 32 cells are read and written as 256-byte runs, the tile is kept in LDS, and
// the mirrored half is written from its transpose, again as 256-byte runs;
// k_norm writes the mirror one element per cache line.
// Grid: (individuals of all windows, tile pairs of the largest window), 256 threads.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_norm_tile(const WinDesc* __restrict__ wins, int n_win,
                                                   const unsigned long long* __restrict__ win_max,
                                                   double* __restrict__ cells, double* __restrict__ full)
{
    __shared__ double tile[32][33];
    int lo = 0, hi = n_win - 1;                      // last window with ind_base <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (wins[mid].ind_base <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WinDesc W = wins[lo];
    const int n = W.n_hap, T = (n + 31) >> 5;
    const int tp = (int)blockIdx.y;
    if (tp >= T * (T + 1) / 2) return;
    const int ta = tri_row_of(tp, T), tb = ta + (int)(tp - tri_row_start(ta, T));
    const int i = (int)(blockIdx.x - W.ind_base);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const double mx = from_ordered_bits(win_max[lo]);
    double* F = full + W.full_base + (int64_t)i * n * W.ld;
    double* C = cells + W.cell_base + (int64_t)i * W.n_gen;
    for (int r = ty; r < 32; r += 8) {
        const int a = ta * 32 + r, b = tb * 32 + tx;
        double p = 0.0;
        if (a < n && b < n && a <= b) {
            const int64_t c = tri_row_start(a, n) + (b - a);
            p = exp(C[c] - mx);                                              // cpopulation.pyx:732-734
            C[c] = p;
            F[(int64_t)a * W.ld + b] = p;
        }
        tile[r][tx] = p;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {               // row b2 of the mirror: elements (a2, b2) of the tile's column r
        const int b2 = tb * 32 + r, a2 = ta * 32 + tx;
        if (a2 < n && b2 < n && a2 < b2) F[(int64_t)b2 * W.ld + a2] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------
// k_em: Caller.call + EMiteration, platypus/cpopulation.pyx:741-890.
//
// With P the symmetric matrix of an individual's normalised genotype
// likelihoods and f the frequencies, the reference's per-genotype terms
//     csr(a,b) = P(a,b) f_a f_b (1 + (a != b)),  csrSum = sum csr     (:847-849)
// and the per-haplotype totals of :869-880 regroup to a matrix-vector product:
//     y = P f,   csrSum = f . y,   total_k = sum_i 2 f_k y_k / csrSum_i ,
// O(n_hap^2) per individual and iteration instead of O(n_hap^3).  One
// wavefront takes an individual at a time (lanes over k, a sequential loop over
// the rows of the full matrix, which adjacent lanes read at adjacent addresses);
// the wavefronts' partial totals are added in wavefront order.
//
// The genotype calls (:773-797) use the csr values of the LAST iteration, i.e.
// the frequencies before its update; they are recomputed from those at the end.
// LDS: f, f_prev, f_new [n_max] | per wavefront: y [n_max], acc [n_max] | staged matrices
// (a window of at most stage_doubles matrix elements is copied there once and read from LDS in every iteration).
// ---------------------------------------------------------------------------
// Columns k0 + c * 64 + lane, c < NC, of y = P f for one wavefront: yv[k] = y_k, returns the
// lane's share of f . y.  16 independent loads are in flight per lane (NC column chunks x
// 16 / NC rows, all loads of a group issued before its sums); every y_k is still summed over
// the rows in row order.
template <int NC, int INFLIGHT = 16>
__device__ __forceinline__ double matvec_pass(const double* P, const double* f_src, double* yv,
                                              int n, int ld, int k0, int lane, int r_begin = 0, int r_end = -1)
{
    if (r_end < 0) r_end = n;
    constexpr int RU = INFLIGHT / NC;
    double y[NC];
    int kk[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { const int k = k0 + c * 64 + lane; kk[c] = k < n ? k : n - 1; y[c] = 0.0; }
    int r = r_begin;
    for (; r + RU <= r_end; r += RU) {
        double pv[RU][NC];
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) pv[u][c] = P[(int64_t)(r + u) * ld + kk[c]];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const double fr = f_src[r + u];
#pragma unroll
            for (int c = 0; c < NC; ++c) y[c] += pv[u][c] * fr;
        }
    }
    for (; r < r_end; ++r) {
        const double fr = f_src[r];
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c] += P[(int64_t)r * ld + kk[c]] * fr;
    }
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = k0 + c * 64 + lane;
        if (k < n) { yv[k] = y[c]; part += f_src[k] * y[c]; }
    }
    return part;
}

template <bool STAGED>
__device__ __forceinline__ void em_window(double* em_lds, double* s_red, int* s_with_data_p,
                                          const WinDesc* __restrict__ wins,
                                          const int32_t* __restrict__ ioff,
                                          const double* __restrict__ full,
                                          int max_iters, int n_max,
                                          double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                          double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                          double* __restrict__ best_val)
{
    const WinDesc W = wins[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_thr = blockDim.x, n_wav = n_thr >> 6;
    const int n = W.n_hap;
    double* f = em_lds;
    double* f_prev = f + n_max;
    double* f_new = f_prev + n_max;
    double* yv = f_new + n_max + (size_t)wave * 2 * n_max;
    double* acc = yv + n_max;
    const int32_t* io = ioff + W.ioff_base;

    // STAGED: the window's matrices are copied to LDS once and every iteration reads them there.
    // The two cases are separate instantiations so that each pointer keeps its address space
    // (one pointer that may be either becomes a flat pointer, and every flat load waits for
    // all memory counters: the loads of the product below would run one at a time).
    double* const stage = em_lds + (size_t)(3 + 2 * n_wav) * n_max;
    const double* const P_glob = full + W.full_base;
    if (STAGED) {
        const int tot = W.n_ind * n * W.ld;
        for (int k = tid; k < tot; k += n_thr) stage[k] = P_glob[k];
    }

    if (tid == 0) {
        int c = 0;
        for (int i = 0; i < W.n_ind; ++i) c += io[i + 1] != io[i];
        *s_with_data_p = c;
    }
    for (int k = tid; k < n; k += n_thr) f[k] = 1.0 / n;                    // :753,763-764
    __syncthreads();
    const int n_with_data = *s_with_data_p;
    double eps = 1.0 / (W.n_ind * 2 * 2);                                    // :751
    eps = eps < 1e-3 ? eps : 1e-3;
    double max_change = eps + 1;
    int iters = 0;

    // y = P_i f_src and csrSum = f_src . y for individual i, by this wavefront: 256 columns
    // per pass, 64 per lane group (see matvec_pass).
    auto matvec = [&](int i, const double* f_src) -> double {
        const double* P = (STAGED ? (const double*)stage : P_glob) + (int64_t)i * n * W.ld;
        double part = 0.0;
        for (int k0 = 0; k0 < n; k0 += 256) {
            const int nc = (n - k0 + 63) >> 6;
            if (nc >= 4)      part += matvec_pass<4>(P, f_src, yv, n, W.ld, k0, lane);
            else if (nc == 3) part += matvec_pass<3>(P, f_src, yv, n, W.ld, k0, lane);
            else if (nc == 2) part += matvec_pass<2>(P, f_src, yv, n, W.ld, k0, lane);
            else              part += matvec_pass<1>(P, f_src, yv, n, W.ld, k0, lane);
        }
        return wave_sum(part);
    };

    while (max_change > eps && iters < max_iters) {                          // :766-768
        for (int k = lane; k < n; k += 64) acc[k] = 0.0;
        for (int i = wave; i < W.n_ind; i += n_wav) {
            if (io[i + 1] == io[i]) continue;                                // :834-835
            const double s = matvec(i, f);
            for (int k = lane; k < n; k += 64) {
                const double t = 2.0 * f[k] * yv[k];
                acc[k] += s > 0.0 ? t / s : t;                               // :852-854
            }
        }
        __syncthreads();
        double ch = 0.0;
        for (int k = tid; k < n; k += n_thr) {
            double t = 0.0;
            for (int w = 0; w < n_wav; ++w) t += em_lds[(size_t)3 * n_max + (size_t)w * 2 * n_max + n_max + k];
            const double nf = t / (2 * n_with_data);                         // :883
            f_new[k] = nf;
            ch = fmax(ch, fabs(f[k] - nf));                                  // :885-888
        }
        ch = wave_max(ch);
        if (lane == 0) s_red[wave] = ch;
        __syncthreads();
        ch = 0.0;
        for (int w = 0; w < n_wav; ++w) ch = fmax(ch, s_red[w]);
        max_change = ch;
        for (int k = tid; k < n; k += n_thr) { f_prev[k] = f[k]; f[k] = f_new[k]; }
        ++iters;
        __syncthreads();
    }

    for (int k = tid; k < n; k += n_thr) freq_out[W.hap_base + k] = f[k];
    if (tid == 0) { iters_out[blockIdx.x] = iters; change_out[blockIdx.x] = max_change; }

    // genotype calls from the last iteration's csr values: first maximum in genotype order
    for (int i = wave; i < W.n_ind; i += n_wav) {
        if (io[i + 1] == io[i]) {                                            // :777-780
            if (lane == 0) { best_gen[W.ind_base + i] = -1; best_val[W.ind_base + i] = 0.0; }
            continue;
        }
        if (iters == 0) {                                                    // csr still all zero (calloc, :701)
            if (lane == 0) { best_gen[W.ind_base + i] = 0; best_val[W.ind_base + i] = 0.0; }
            continue;
        }
        const double s = matvec(i, f_prev);
        const double* P = (STAGED ? (const double*)stage : P_glob) + (int64_t)i * n * W.ld;
        double bv = 0.0;
        int bj = 0x7fffffff;
        for (int a = 0; a < n; ++a) {
            const int64_t row = tri_row_start(a, n);
            for (int b = a + lane; b < n; b += 64) {
                double v = P[(int64_t)a * W.ld + b] * f_prev[a] * f_prev[b] * (1 + (a != b));   // :847
                if (s > 0.0) v /= s;
                const int j = (int)(row + b - a);
                if (bj == 0x7fffffff || v > bv) { bv = v; bj = j; }          // :791 within the lane: j ascending
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oj = __shfl_xor(bj, off);
            if (oj != 0x7fffffff && (bj == 0x7fffffff || ov > bv || (ov == bv && oj < bj))) { bv = ov; bj = oj; }
        }
        if (lane == 0) { best_gen[W.ind_base + i] = bj; best_val[W.ind_base + i] = bv; }
    }
}

__global__ __launch_bounds__(1024) void k_em(const WinDesc* __restrict__ wins,
                                             const int32_t* __restrict__ ioff,
                                             const double* __restrict__ full,
                                             int max_iters, int n_max, int staged, int skip_small,
                                             double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                             double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                             double* __restrict__ best_val)
{
    extern __shared__ double em_lds[];
    __shared__ double s_red[16];
    __shared__ int s_with_data;
    const WinDesc W0 = wins[blockIdx.x];           // windows whose matrices fit the staging area keep them in LDS
    if (skip_small && em_is_exact(W0.n_hap, W0.n_ind, W0.n_gen)) return;     // k_em_exact's windows
    if ((int64_t)W0.n_ind * W0.n_hap * W0.n_hap <= (int64_t)staged)
        em_window<true>(em_lds, s_red, &s_with_data, wins, ioff, full, max_iters, n_max,
                        freq_out, iters_out, change_out, best_gen, best_val);
    else
        em_window<false>(em_lds, s_red, &s_with_data, wins, ioff, full, max_iters, n_max,
                         freq_out, iters_out, change_out, best_gen, best_val);
}

// ---------------------------------------------------------------------------
// The same EM for windows whose matrices do not fit the LDS staging area of k_em
// (one workgroup per window would wait on global-memory latency, and leave most
// of the chip idle when the windows are few): every iteration is a pair of launches.
//   k_em_step    grid (window, individual): the workgroup's wavefronts split the
//                rows of y = P_i f and store the individual's share
//                2 f_k y_k / csrSum_i of every haplotype's total
//   k_em_update  grid (window): adds the shares in individual order, new
//                frequencies, largest change, convergence flag
//   k_em_calls   grid (window, individual): genotype calls from the last
//                iteration's old frequencies
// Frequencies live in global memory between launches; a window that has
// converged (or reached max_iters) makes its workgroups return at once, and
// `active` counts the windows still running so that the host can stop launching.
// ---------------------------------------------------------------------------
struct EmState {
    double* f;          // per haplotype: current frequencies (the final output)
    double* f_prev;     // per haplotype: frequencies before the last update
    double* share;      // per (individual, haplotype) of every window
    int32_t* iters;     // per window
    double* change;     // per window
    int32_t* done;      // per window
    int32_t* active;    // one counter
    int32_t* arrived;   // per window: workgroups of the running step that have stored their share
    double* csr;        // per (window, individual): csrSum of the individual's last step (k_em_calls divides by it)
    const int64_t* share_base;   // per window: offset into share
};

// skip_small: the windows em_is_exact names belong to k_em_exact; they count as done from the start
// (n_running = the number of the others)
__global__ __launch_bounds__(256) void k_em_init(const WinDesc* __restrict__ wins, int n_win, EmState S, int skip_small,
                                                 int n_running)
{
    const int w = blockIdx.x;
    const WinDesc W = wins[w];
    if (skip_small && em_is_exact(W.n_hap, W.n_ind, W.n_gen)) {
        if (threadIdx.x == 0) {
            S.done[w] = 1; S.arrived[w] = 0; S.iters[w] = 0;
            if (w == 0) *S.active = n_running;
        }
        return;
    }
    for (int k = threadIdx.x; k < W.n_hap; k += 256) {
        S.f[W.hap_base + k] = 1.0 / W.n_hap;                                 // cpopulation.pyx:753,763-764
        S.f_prev[W.hap_base + k] = 1.0 / W.n_hap;
    }
    if (threadIdx.x == 0) {
        double eps = 1.0 / (W.n_ind * 2 * 2);
        eps = eps < 1e-3 ? eps : 1e-3;
        S.iters[w] = 0; S.change[w] = eps + 1; S.done[w] = 0; S.arrived[w] = 0;   // cpopulation.pyx:751-752
        if (w == 0) *S.active = skip_small ? n_running : n_win;
    }
}

// One EM step for ONE individual per workgroup (grid: window x individual): the wavefronts
// (four or eight) take an equal share of the matrix rows each and all columns, their partial products are added in
// wavefront order.  Several times the wavefronts of one wavefront per individual, i.e. several times the bytes in
// flight: the last iterations of a batch, when few windows are still running, are bound by the
// memory latency of the few wavefronts left, not by bandwidth.
__global__ __launch_bounds__(512) void k_em_step(const WinDesc* __restrict__ wins, const int32_t* __restrict__ ioff,
                                                 const double* __restrict__ full, EmState S, int n_max)
{
    extern __shared__ double em_lds[];
    __shared__ double s_part[8];
    const int w = blockIdx.x;
    if (S.done[w]) return;
    const WinDesc W = wins[w];
    const int i = blockIdx.y;
    if (i >= W.n_ind) return;
    const int32_t* io = ioff + W.ioff_base;
    if (io[i + 1] == io[i]) return;                                          // cpopulation.pyx:834-835
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_thr = blockDim.x, n_wav = n_thr >> 6;
    const int n = W.n_hap;
    double* f = em_lds;
    double* yv = em_lds + n_max + (size_t)wave * n_max;
    const double* f_glob = S.f + W.hap_base;
    for (int k = tid; k < n; k += n_thr) f[k] = f_glob[k];
    __syncthreads();
    const double* P = full + W.full_base + (int64_t)i * n * W.ld;
    const int rows = (n + n_wav - 1) / n_wav;
    const int r0 = wave * rows < n ? wave * rows : n, r1 = r0 + rows < n ? r0 + rows : n;
    for (int k0 = 0; k0 < n; k0 += 256) {
        const int nc = (n - k0 + 63) >> 6;
        if (nc >= 4)      matvec_pass<4>(P, f, yv, n, W.ld, k0, lane, r0, r1);
        else if (nc == 3) matvec_pass<3>(P, f, yv, n, W.ld, k0, lane, r0, r1);
        else if (nc == 2) matvec_pass<2>(P, f, yv, n, W.ld, k0, lane, r0, r1);
        else              matvec_pass<1>(P, f, yv, n, W.ld, k0, lane, r0, r1);
    }
    __syncthreads();
    double part = 0.0;
    double* y0 = em_lds + n_max;
    for (int k = tid; k < n; k += n_thr) {
        double y = y0[k];
        for (int v = 1; v < n_wav; ++v) y += y0[(size_t)v * n_max + k];      // wavefront order
        y0[k] = y;                                    // this thread is the only reader of column k
        part += f[k] * y;
    }
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    double s = s_part[0];                                                    // csrSum, cpopulation.pyx:849
    for (int v = 1; v < n_wav; ++v) s += s_part[v];
    if (tid == 0) S.csr[W.ind_base + i] = s;
    double* sh = S.share + S.share_base[w] + (int64_t)i * n;
    for (int k = tid; k < n; k += n_thr) {
        const double t = 2.0 * f[k] * y0[k];
        sh[k] = s > 0.0 ? t / s : t;                                         // cpopulation.pyx:852-854
    }
}

// Shares written and read within ONE launch (by different workgroups, possibly on different dies of the chip)
// go past the caches that are not kept coherent between dies: device-scope stores and loads, so that no
// workgroup has to write back or invalidate a whole cache to publish or to see them.
__device__ __forceinline__ void share_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SAME_LAUNCH>
__device__ __forceinline__ double share_load(const double* p) {
    if (SAME_LAUNCH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// New frequencies of one window from the individuals' shares (added in individual order), largest change,
// convergence flag: the work of one workgroup of n_thr threads (at most 16 wavefronts).  s_has: one byte per
// individual in LDS.
#ifndef PG_UPDATE_BATCH
#define PG_UPDATE_BATCH 16
#endif
constexpr int kUpdateBatch = PG_UPDATE_BATCH;
template <bool SAME_LAUNCH>
__device__ __forceinline__ void em_update_window(const WinDesc& W, const int32_t* __restrict__ io, EmState S, int w,
                                                 int max_iters, int tid, int n_thr, unsigned char* s_has,
                                                 double* s_red, int* s_count)
{
    const int n = W.n_hap, n_wav = n_thr >> 6;
    // which individuals have reads: read once by the whole workgroup (every thread reading all of them
    // was most of this kernel's time)
    int mine = 0;
    for (int i = tid; i < W.n_ind; i += n_thr) {
        const int h = io[i + 1] != io[i];
        s_has[i] = (unsigned char)h;
        mine += h;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((tid & 63) == 0) s_count[tid >> 6] = mine;
    __syncthreads();
    int n_with_data = 0;
    for (int v = 0; v < n_wav; ++v) n_with_data += s_count[v];
    const double* sh = S.share + S.share_base[w];
    double ch = 0.0;
    for (int k = tid; k < n; k += n_thr) {
        double t = 0.0;
        // kUpdateBatch shares are fetched before the first is added (the compiler keeps loads where they
        // are written, and fifty dependent load-add steps are fifty memory latencies); the
        // additions stay in individual order
        for (int i0 = 0; i0 < W.n_ind; i0 += kUpdateBatch) {
            double v[kUpdateBatch];
#pragma unroll
            for (int j = 0; j < kUpdateBatch; ++j) v[j] = i0 + j < W.n_ind ? share_load<SAME_LAUNCH>(sh + (int64_t)(i0 + j) * n + k) : 0.0;
#pragma unroll
            for (int j = 0; j < kUpdateBatch; ++j)
                if (i0 + j < W.n_ind && s_has[i0 + j]) t += v[j];             // cpopulation.pyx:869-880, individual order
        }
        const double old = S.f[W.hap_base + k], nf = t / (2 * n_with_data);  // :883
        S.f_prev[W.hap_base + k] = old;
        S.f[W.hap_base + k] = nf;
        ch = fmax(ch, fabs(old - nf));                                       // :885-888
    }
    ch = wave_max(ch);
    if ((tid & 63) == 0) s_red[tid >> 6] = ch;
    __syncthreads();
    if (tid == 0) {
        for (int v = 1; v < n_wav; ++v) ch = fmax(ch, s_red[v]);
        double eps = 1.0 / (W.n_ind * 2 * 2);
        eps = eps < 1e-3 ? eps : 1e-3;
        const int it = S.iters[w] + 1;
        S.iters[w] = it;
        S.change[w] = ch;
        if (!(ch > eps && it < max_iters)) {                                 // :766
            S.done[w] = 1;
            atomicSub(S.active, 1);
        }
    }
}

// The same step from the upper triangle alone: the matrix is symmetric, so every element P_ab (a <= b) is
// fetched once and used twice, y_b += P_ab f_a and (b > a) y_a += P_ab f_b -- half the bytes of k_em_step, which
// streams both halves.  A wavefront takes blocks of kSymRows rows and walks each from the diagonal to the right edge
// in tiles of 64 columns: each lane owns a column (kSymRows independent loads in flight, and the next tile's are
// issued before this tile's are used), adds the column part in registers and keeps one row-part sum per row of
// its own; at the end of a row block those are handed to eight lanes per row through an 8 x 64 LDS tile --
// a transposition, no cross-lane reduction per row beyond three exchange steps.  Every wavefront accumulates
// into its own copy of y (no atomics: sums are reproducible run to run); the copies are added in wavefront order.
// Row blocks are dealt back and forth (0..W-1, 2W-1..W, ...): a block near the top has the most tiles.
// LDS: f [n_pad] | per wavefront: y [n_pad], tile [8][65]; n_pad = the largest sym_pad of the batch's windows.
constexpr int kSymRows = 8;
constexpr int kSymTile = 8 * 65;
constexpr int kSymEdge = 8;           // columns past whole wavefronts that are read as rows of the other half

struct SymTile { int a0, cb; bool ok; };

// the tile after `t` on this wavefront's walk; `round` counts its row blocks
__device__ __forceinline__ SymTile sym_next(SymTile t, int& round, int wave, int n_wav, int n, int n_cb)
{
    ++t.cb;
    if (t.ok && t.cb < n_cb) return t;
    ++round;
    t.a0 = kSymRows * (round * n_wav + ((round & 1) ? n_wav - 1 - wave : wave));
    t.cb = t.a0 >> 6;
    t.ok = t.a0 < n;
    return t;
}

// No load is conditional (the compiler would wait for all of them before the first use).  Lanes past the
// matrix edge re-read the last column and lanes left of the diagonal re-read the diagonal -- lines the row
// fetches anyway -- and rows past the edge read whatever follows (the next matrix, the zeroed slack behind the
// last one); what they fetch meets a frequency of zero from the padded copy in LDS, or is discarded.
// ld: row length of the matrix; lim: rows and columns the tiles cover (see sym_share).
template <bool DIAG>
__device__ __forceinline__ void sym_load(double (&p)[kSymRows], const double* __restrict__ P, SymTile t, int ld, int lim, int lane)
{
    const int b = min((t.cb << 6) + lane, lim - 1);
    const double* row = P + (int64_t)t.a0 * ld;
#pragma unroll
    for (int r = 0; r < kSymRows; ++r) p[r] = row[(int64_t)r * ld + (DIAG ? max(b, min(t.a0 + r, lim - 1)) : b)];
}

__device__ __forceinline__ void sym_load_any(double (&p)[kSymRows], const double* __restrict__ P, SymTile t, int ld, int lim, int lane)
{
    if (t.cb == t.a0 >> 6) sym_load<true>(p, P, t, ld, lim, lane);
    else                   sym_load<false>(p, P, t, ld, lim, lane);
}

// One tile's contribution: the column part goes to the wavefront's copy of y, the row part to the lane's sums z.
// DIAG: the tile holds diagonal elements, and what the lanes left of them fetched is not theirs.
template <bool DIAG>
__device__ __forceinline__ void sym_tile(const double (&p)[kSymRows], double (&z)[kSymRows], SymTile t, int lane,
                                         const double* f, double* yw)
{
    const int a0 = t.a0, b = (t.cb << 6) + lane;
    const double fb = f[b];
    double yc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < kSymRows; ++r) {
        const int a = a0 + r;
        const double v = p[r];
        yc[r & 3] += (!DIAG || b >= a ? v : 0.0) * f[a];
        z[r] += (!DIAG || b > a ? v : 0.0) * fb;
    }
    yw[b] += (yc[0] + yc[1]) + (yc[2] + yc[3]);
}

// End of a row block: the lanes' row-part sums, added over the wavefront, go to y; z is cleared.
__device__ __forceinline__ void sym_rows(double (&z)[kSymRows], int a0, int lane, double* yw, double* tile)
{
    const int q = lane & 7, rr = lane >> 3;
#pragma unroll
    for (int half = 0; half < kSymRows / 8; ++half) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { tile[r * 65 + lane] = z[8 * half + r]; z[8 * half + r] = 0.0; }
        double s0 = 0.0, s1 = 0.0;                       // LDS operations of one wavefront complete in order
#pragma unroll
        for (int j = 0; j < 8; j += 2) { s0 += tile[rr * 65 + 8 * q + j]; s1 += tile[rr * 65 + 8 * q + j + 1]; }
        double srow = s0 + s1;
        srow += dpp_f64<0xB1>(srow);                     // quad_perm [1,0,3,2]
        srow += dpp_f64<0x4E>(srow);                     // quad_perm [2,3,0,1]
        srow += dpp_f64<0x141>(srow);                    // row_half_mirror: the other quad of the eight
        if (q == 0) yw[a0 + 8 * half + rr] += srow;
    }
}

__device__ __forceinline__ void sym_step(const double (&p)[kSymRows], double (&z)[kSymRows], SymTile t, bool last,
                                         int lane, const double* f, double* yw, double* tile)
{
    if (t.cb == t.a0 >> 6) sym_tile<true>(p, z, t, lane, f, yw);
    else                   sym_tile<false>(p, z, t, lane, f, yw);
    if (last) sym_rows(z, t.a0, lane, yw, tile);
}

// columns of an n-wide matrix that the step reads as rows of the other half (sym_share)
__host__ __device__ inline int sym_edge(int n) { return (n >= 64 && (n & 63) <= kSymEdge) ? (n & 63) : 0; }
// LDS doubles per vector for an n-wide matrix: the tiles' columns rounded up to whole wavefronts (their rows, to
// whole row blocks, are fewer), and the matrix's own width
__host__ __device__ inline int sym_pad(int n)
{
    const int tiles = (n - sym_edge(n) + 63) & ~63, own = (n + 7) & ~7;
    return tiles > own ? tiles : own;
}
// doubles a step may read past the end of the last matrix
__host__ __device__ inline int64_t sym_slack(int n_max) { return (int64_t)kSymRows * (n_max + 16 + 64); }

// One individual's share of every haplotype's total, from the upper triangle of its matrix.
__device__ __forceinline__ void sym_share(const WinDesc& W, int i, const double* __restrict__ full, EmState S, int w,
                                          int n_pad, double* em_lds, double* s_part)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               // the tile walk stays in scalar registers
    const int n_thr = blockDim.x, n_wav = n_thr >> 6;
    const int n = W.n_hap;
    double* f = em_lds;
    double* yw = em_lds + n_pad + (size_t)wave * n_pad;
    double* tile = em_lds + (size_t)(1 + n_wav) * n_pad + (size_t)wave * kSymTile;
    const double* P = full + W.full_base + (int64_t)i * n * W.ld;
    // A matrix a few columns wider than whole wavefronts would pay a tile per row block for those columns.
    // They are taken from the other half instead: column b of the upper triangle is row b of the lower one,
    // contiguous in memory.  The tiles then cover the leading lim x lim block, the last n - lim rows are read whole.
    const int rem = sym_edge(n);
    const int lim = n - rem;
    const int n_cb = (lim + 63) >> 6;
    int round = -1;
    SymTile t0 = sym_next(SymTile{0, 0, false}, round, wave, n_wav, lim, n_cb), t1;
    double p0[kSymRows], p1[kSymRows], z[kSymRows];
#pragma unroll
    for (int r = 0; r < kSymRows; ++r) z[r] = 0.0;
    if (t0.ok) sym_load_any(p0, P, t0, W.ld, lim, lane);    // in flight while the frequencies are staged
    const double* f_glob = S.f + W.hap_base;
    for (int k = tid; k < n_pad; k += n_thr) f[k] = k < n ? f_glob[k] : 0.0;
    for (int k = lane; k < n_pad; k += 64) yw[k] = 0.0;
    __syncthreads();
    // two register sets take turns: the next tile's loads are issued before this tile's values are used
    while (t0.ok) {
        t1 = sym_next(t0, round, wave, n_wav, lim, n_cb);
        sym_load_any(p1, P, t1.ok ? t1 : t0, W.ld, lim, lane);
        sym_step(p0, z, t0, !(t1.ok && t1.a0 == t0.a0), lane, f, yw, tile);
        if (!t1.ok) break;
        t0 = sym_next(t1, round, wave, n_wav, lim, n_cb);
        sym_load_any(p0, P, t0.ok ? t0 : t1, W.ld, lim, lane);
        sym_step(p1, z, t1, !(t0.ok && t0.a0 == t1.a0), lane, f, yw, tile);
    }
    // the last rows, one per wavefront in turn (the last wavefronts first: they have walked the fewest tiles):
    // element (b, a), a <= b, is P_ab
    for (int j = 0; j < rem; ++j) {
        if ((n_wav - 1 - j % n_wav) != wave) continue;
        const int b = lim + j;
        const double* row = P + (int64_t)b * W.ld;
        const double fb = f[b];
        double acc = 0.0;
        for (int a0 = 0; a0 <= b; a0 += 64) {
            const int a = a0 + lane;
            const double v = a <= b ? row[a] : 0.0;
            acc += v * f[min(a, n_pad - 1)];                                 // y_b += P_ab f_a, the diagonal included
            if (a < b) yw[a] += v * fb;                                      // y_a += P_ab f_b
        }
        acc = wave_sum(acc);
        if (lane == 0) yw[b] += acc;
    }
    __syncthreads();
    double part = 0.0;
    double* y0 = em_lds + n_pad;
    for (int k = tid; k < n; k += n_thr) {
        double y = y0[k];
        for (int v = 1; v < n_wav; ++v) y += y0[(size_t)v * n_pad + k];      // wavefront order
        y0[k] = y;                                    // this thread is the only reader of column k
        part += f[k] * y;
    }
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    double s = s_part[0];                                                    // csrSum, cpopulation.pyx:849
    for (int v = 1; v < n_wav; ++v) s += s_part[v];
    if (tid == 0) S.csr[W.ind_base + i] = s;
    double* sh = S.share + S.share_base[w] + (int64_t)i * n;
    for (int k = tid; k < n; k += n_thr) {
        const double tt = 2.0 * f[k] * y0[k];
        share_store(sh + k, s > 0.0 ? tt / s : tt);                          // cpopulation.pyx:852-854
    }
}

// grid (window, individual).  Dynamic LDS: the layout above (n_pad from the host), and at least one byte per
// individual of a window.
__global__ __launch_bounds__(1024) void k_em_step_sym(const WinDesc* __restrict__ wins, const int32_t* __restrict__ ioff,
                                                      const double* __restrict__ full, EmState S, int n_pad,
                                                      int max_iters)
{
    extern __shared__ double em_lds[];
    __shared__ double s_part[16];
    __shared__ int s_count[16];
    __shared__ int s_last;
    const int w = blockIdx.x;
    if (S.done[w]) return;
    const WinDesc W = wins[w];
    const int i = blockIdx.y;
    if (i >= W.n_ind) return;
    const int32_t* io = ioff + W.ioff_base;
    const int tid = threadIdx.x;
    if (io[i + 1] != io[i])                                                  // cpopulation.pyx:834-835
        sym_share(W, i, full, S, w, n_pad, em_lds, s_part);
    // The window's last workgroup to get here does the window's update: every workgroup's shares have reached
    // device scope before it counts itself (its wavefronts wait for their stores, then the barrier), the
    // frequencies that are replaced have been read by all of them, and nothing reads the new ones before the
    // next launch.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
        const int last = __hip_atomic_fetch_add(&S.arrived[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == W.n_ind - 1;
        if (last) S.arrived[w] = 0;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    em_update_window<true>(W, io, S, w, max_iters, tid, blockDim.x, reinterpret_cast<unsigned char*>(em_lds), s_part, s_count);
}

// Genotype calls (cpopulation.pyx:773-797) from the last iteration's old frequencies: one workgroup per
// (window, individual) scans the upper triangle of the individual's matrix once, its rows split between the
// wavefronts in ranges of about equal size; csrSum is the one the individual's last step left in S.csr (the
// step of the frequencies now in f_prev); the wavefronts' best genotypes are compared in wavefront order, i.e.
// in genotype order (first maximum wins).
__global__ __launch_bounds__(512) void k_em_calls(const WinDesc* __restrict__ wins, const int32_t* __restrict__ ioff,
                                                  const double* __restrict__ full, EmState S, int n_max, int skip_small,
                                                  int32_t* __restrict__ best_gen, double* __restrict__ best_val)
{
    extern __shared__ double em_lds[];
    __shared__ double s_bv[8];
    __shared__ int s_bj[8];
    const int w = blockIdx.x;
    const WinDesc W = wins[w];
    const int i = blockIdx.y;
    if (i >= W.n_ind) return;
    if (skip_small && em_is_exact(W.n_hap, W.n_ind, W.n_gen)) return;        // k_em_exact's windows
    const int32_t* io = ioff + W.ioff_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_thr = blockDim.x, n_wav = n_thr >> 6;
    if (io[i + 1] == io[i]) {                                                // cpopulation.pyx:777-780
        if (tid == 0) { best_gen[W.ind_base + i] = -1; best_val[W.ind_base + i] = 0.0; }
        return;
    }
    if (S.iters[w] == 0) {                                                   // csr still all zero (calloc, :701)
        if (tid == 0) { best_gen[W.ind_base + i] = 0; best_val[W.ind_base + i] = 0.0; }
        return;
    }
    const int n = W.n_hap;
    double* f = em_lds;
    const double* f_glob = S.f_prev + W.hap_base;
    for (int k = tid; k < n; k += n_thr) f[k] = f_glob[k];
    __syncthreads();
    const double* P = full + W.full_base + (int64_t)i * n * W.ld;
    // rows of the upper triangle dealt to the wavefronts as contiguous ranges of about equal element counts
    // (ascending: a later wavefront holds later genotypes, which the tie rule below relies on)
    const int64_t total = (int64_t)n * (n + 1) / 2, per = (total + n_wav - 1) / n_wav;
    int r0 = n, r1 = n;
    int64_t before = 0;                                                      // elements of the rows above a
    const int64_t mine_lo = (int64_t)wave * per, mine_hi = mine_lo + per;
    for (int a = 0; a < n; ++a) {                                            // a few hundred rows: every thread walks them
        if (before >= mine_lo && before < mine_hi) {                         // before / per == wave, without the division
            if (r0 == n) r0 = a;
            r1 = a + 1;
        }
        before += n - a;
    }
    // csrSum of the individual's last step, i.e. of the frequencies in f_prev: the step kernel left it here
    const double s = S.csr[W.ind_base + i];
    double bv = 0.0;
    int bj = 0x7fffffff;
    // The wavefront's pieces -- 64 consecutive elements of a row, from its diagonal on -- eight at a time: the eight
    // loads are issued before the first value is used (one piece per step is one memory latency per step).  No load
    // is conditional: a piece past the range re-reads the last row, a lane past the row's end the row's last element.
    // A lane meets its genotypes in ascending order, so "larger wins" keeps the first of equal maxima.
    int a_s = __builtin_amdgcn_readfirstlane(r0), c_s = 0;                   // the piece walk stays in scalar registers
    const int r1_s = __builtin_amdgcn_readfirstlane(r1);
    while (a_s < r1_s) {
        double p[8];
        int pa[8], pb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = min(a_s, r1_s - 1);
            pa[u] = a_s < r1_s ? a : -1;
            pb[u] = a + c_s * 64 + lane;
            p[u] = P[(int64_t)a * W.ld + min(pb[u], n - 1)];
            ++c_s;
            if (a_s + c_s * 64 >= n) { ++a_s; c_s = 0; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = pa[u], b = pb[u];
            if (a < 0 || b >= n) continue;
            double v = p[u] * f[a] * f[b] * (1 + (a != b));                  // cpopulation.pyx:847
            if (s > 0.0) v /= s;
            const int j = (int)(tri_row_start(a, n) + b - a);
            if (bj == 0x7fffffff || v > bv) { bv = v; bj = j; }             // cpopulation.pyx:791
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oj = __shfl_xor(bj, off);
        if (oj != 0x7fffffff && (bj == 0x7fffffff || ov > bv || (ov == bv && oj < bj))) { bv = ov; bj = oj; }
    }
    if (lane == 0) { s_bv[wave] = bv; s_bj[wave] = bj; }
    __syncthreads();
    if (tid == 0) {
        for (int v = 1; v < n_wav; ++v)
            if (s_bj[v] != 0x7fffffff && (bj == 0x7fffffff || s_bv[v] > bv)) { bv = s_bv[v]; bj = s_bj[v]; }
        best_gen[W.ind_base + i] = bj; best_val[W.ind_base + i] = bv;
    }
}

__global__ __launch_bounds__(256) void k_em_update(const WinDesc* __restrict__ wins, const int32_t* __restrict__ ioff,
                                                   EmState S, int max_iters)
{
    __shared__ double s_red[4];
    __shared__ int s_count[4];
    extern __shared__ unsigned char s_has[];             // per individual: it has reads
    const int w = blockIdx.x;
    if (S.done[w]) return;
    const WinDesc W = wins[w];
    em_update_window<false>(W, ioff + W.ioff_base, S, w, max_iters, threadIdx.x, 256, s_has, s_red, s_count);
}

// ---------------------------------------------------------------------------
// k_em_small: the same computation for windows of at most 8 haplotypes (the
// usual window has 2-4), one wavefront per window, everything in registers.
// A matrix has n^2 <= 64 elements.  With several individuals and n <= 5,
// G = 64 / n^2 >= 2 individuals are taken side by side, lane l of slot
// g = l / n^2 owning element (r, k) = (e / n, e % n), e = l % n^2, of individual
// s * G + g in step s; otherwise one individual at a time, lane = element.  The
// product sums its rows in row order through lane reads, the frequencies live
// replicated in the lanes, and the individuals' shares are added to the totals
// in individual order.  Every sum has the shape it has in k_em with one
// individual per wavefront, so the kernels return the same doubles.
// ---------------------------------------------------------------------------
// One individual at a time: lane l < n^2 owns the matrix element (r, k) = (l / n, l % n) (windows of 6 to 8
// haplotypes, whose matrices fill most of a wavefront, and windows with one individual).
__device__ __forceinline__ void em_small_one(const WinDesc& W, int w, int lane,
                                             const int32_t* __restrict__ ioff,
                                             const double* __restrict__ full,
                                             int max_iters,
                                             double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                             double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                             double* __restrict__ best_val)
{
    const int n = W.n_hap, nn = n * n;
    const int32_t* io = ioff + W.ioff_base;
    const bool live = lane < nn;
    const int r = live ? lane / n : 0, k = live ? lane % n : 0;
    const double* Pw = full + W.full_base;
    int n_with_data = 0;
    for (int i = 0; i < W.n_ind; ++i) n_with_data += io[i + 1] != io[i];
    double eps = 1.0 / (W.n_ind * 2 * 2);                                    // cpopulation.pyx:751
    eps = eps < 1e-3 ? eps : 1e-3;
    double max_change = eps + 1;
    int iters = 0;
    double fk = 1.0 / n, fr = 1.0 / n, fk_prev = fk, fr_prev = fr;           // f[k], f[r] of this lane
    const double P0 = live ? Pw[lane] : 0.0;                                  // individual 0 stays in a register

    // y_k (valid in lanes with r == 0, i.e. lanes 0..n-1) and csrSum for the individual whose element is p
    auto product = [&](double p, double f_r, double f_k, double& y) -> double {
        const double t = p * f_r;
        y = 0.0;
        for (int rr = 0; rr < n; ++rr) y += __shfl(t, rr * n + k);          // rows in row order
        return wave_sum((live && r == 0) ? f_k * y : 0.0);
    };
    auto matvec = [&](int i, double f_r, double f_k, double& y) -> double {
        return product(i == 0 ? P0 : (live ? Pw[(int64_t)i * nn + lane] : 0.0), f_r, f_k, y);
    };

    while (max_change > eps && iters < max_iters) {                          // :766-768
        double acc = 0.0;
        // two individuals per pass: their sums are independent chains of lane reads, reductions and a division
        // each, and one wavefront waits for its own instructions' latency -- side by side they overlap (four
        // per pass were slower again); the shares are still added in individual order
        double pa = P0, pb = (live && W.n_ind > 1) ? Pw[nn + lane] : 0.0;
        for (int i = 0; i < W.n_ind; i += 2) {
            const double p0 = pa, p1 = pb;
            if (i + 2 < W.n_ind) pa = live ? Pw[(int64_t)(i + 2) * nn + lane] : 0.0;   // in flight during this pass
            if (i + 3 < W.n_ind) pb = live ? Pw[(int64_t)(i + 3) * nn + lane] : 0.0;
            const bool has0 = io[i + 1] != io[i];                            // :834-835
            double y0, y1;
            if (i + 1 < W.n_ind) {
                const bool has1 = io[i + 2] != io[i + 1];
                const double s0 = product(p0, fr, fk, y0);
                const double s1 = product(p1, fr, fk, y1);
                const double t0 = 2.0 * fk * y0, t1 = 2.0 * fk * y1;
                if (has0) acc += s0 > 0.0 ? t0 / s0 : t0;                    // :852-854
                if (has1) acc += s1 > 0.0 ? t1 / s1 : t1;
            } else if (has0) {
                const double s0 = product(p0, fr, fk, y0);
                const double t0 = 2.0 * fk * y0;
                acc += s0 > 0.0 ? t0 / s0 : t0;
            }
        }
        const double nf = (0.0 + acc) / (2 * n_with_data);                   // :883
        const double ch = wave_max((live && r == 0) ? fabs(fk - nf) : 0.0);  // :885-888
        max_change = ch;
        fk_prev = fk; fr_prev = fr;
        // lanes 0..n-1 hold the new f_k for k = lane; every lane picks up its own k and r
        fk = __shfl(nf, k);
        fr = __shfl(nf, r);
        ++iters;
    }
    if (lane < n) freq_out[W.hap_base + lane] = fk;
    if (lane == 0) { iters_out[w] = iters; change_out[w] = max_change; }

    for (int i = 0; i < W.n_ind; ++i) {                                      // :773-797
        if (io[i + 1] == io[i]) {
            if (lane == 0) { best_gen[W.ind_base + i] = -1; best_val[W.ind_base + i] = 0.0; }
            continue;
        }
        if (iters == 0) {
            if (lane == 0) { best_gen[W.ind_base + i] = 0; best_val[W.ind_base + i] = 0.0; }
            continue;
        }
        double y;
        const double s = matvec(i, fr_prev, fk_prev, y);
        const double p = i == 0 ? P0 : (live ? Pw[(int64_t)i * nn + lane] : 0.0);
        double bv = 0.0;
        int bj = 0x7fffffff;
        if (live && r <= k) {                                                // genotype (a, b) = (r, k)
            double v = p * fr_prev * fk_prev * (1 + (r != k));               // :847
            if (s > 0.0) v /= s;
            bv = v;
            bj = (int)(tri_row_start(r, n) + k - r);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oj = __shfl_xor(bj, off);
            if (oj != 0x7fffffff && (bj == 0x7fffffff || ov > bv || (ov == bv && oj < bj))) { bv = ov; bj = oj; }
        }
        if (lane == 0) { best_gen[W.ind_base + i] = bj; best_val[W.ind_base + i] = bv; }
    }
}

// G = 64 / n^2 >= 2 individuals side by side (windows of up to 5 haplotypes with several individuals).
__device__ __forceinline__ void em_small_slots(const WinDesc& W, int w, int lane,
                                               const int32_t* __restrict__ ioff,
                                               const double* __restrict__ full,
                                               int max_iters,
                                               double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                               double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                               double* __restrict__ best_val)
{
    const int n = W.n_hap, nn = n * n;
    const int G = 64 / nn, steps = (W.n_ind + G - 1) / G;
    const int32_t* io = ioff + W.ioff_base;
    const bool live = lane < G * nn;
    const int slot = live ? lane / nn : 0, sb = slot * nn, e = live ? lane - sb : 0;
    const int r = e / n, k = e % n;
    const double* Pw = full + W.full_base;
    int n_with_data = 0;
    for (int i = 0; i < W.n_ind; ++i) n_with_data += io[i + 1] != io[i];
    double eps = 1.0 / (W.n_ind * 2 * 2);                                    // cpopulation.pyx:751
    eps = eps < 1e-3 ? eps : 1e-3;
    double max_change = eps + 1;
    int iters = 0;
    double fk = 1.0 / n, fr = 1.0 / n, fk_prev = fk, fr_prev = fr;           // f[k], f[r] of this lane

    // element of the slot's individual in step s (zero past the last individual), and whether it has reads
    auto element = [&](int s) -> double {
        const int i = s * G + slot;
        return (live && i < W.n_ind) ? Pw[(int64_t)i * nn + e] : 0.0;
    };
    auto with_data = [&](int s) -> int {
        const int i = s * G + slot;
        return (live && i < W.n_ind) ? (int)(io[i + 1] != io[i]) : 0;
    };
    const double P0 = element(0);                                             // step 0 stays in a register
    const int D0 = with_data(0);

    // y_k (valid in the lanes with r == 0 of every slot) and csrSum of every slot's individual.  The sum over
    // the slot's n lanes has the shape of a wavefront sum over lanes 0..7 (pairs, fours, eight).
    auto matvec = [&](double p, double f_r, double f_k, double& y) -> double {
        const double t = p * f_r;
        y = 0.0;
        for (int rr = 0; rr < n; ++rr) y += __shfl(t, sb + rr * n + k);     // rows in row order
        const double val = (live && r == 0) ? f_k * y : 0.0;
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double got = __shfl(val, sb + (j < n ? j : 0));
            v[j] = j < n ? got : 0.0;
        }
        return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    };

    while (max_change > eps && iters < max_iters) {                          // :766-768
        double acc = 0.0;                                                    // total of this lane's haplotype k
        double p_next = P0;
        int d_next = D0;
        for (int s = 0; s < steps; ++s) {
            const double p = p_next;
            const int has = d_next;
            if (s + 1 < steps) { p_next = element(s + 1); d_next = with_data(s + 1); }   // in flight during this step
            double y;
            const double sum = matvec(p, fr, fk, y);
            const double t = 2.0 * fk * y;
            const double share = sum > 0.0 ? t / sum : t;                    // :852-854, in the lanes with r == 0
            for (int g = 0; g < G && s * G + g < W.n_ind; ++g) {             // the slots in individual order
                const double sh = __shfl(share, g * nn + k);
                if (__shfl(has, g * nn)) acc += sh;                          // :834-835: no reads, no term
            }
        }
        const double nf = (0.0 + acc) / (2 * n_with_data);                   // :883
        const double ch = wave_max(lane < n ? fabs(fk - nf) : 0.0);          // :885-888 (lanes 0..n-1: every k once)
        max_change = ch;
        fk_prev = fk; fr_prev = fr;
        fk = nf;
        fr = __shfl(nf, r);                                                  // lane r holds haplotype r's
        ++iters;
    }
    if (lane < n) freq_out[W.hap_base + lane] = fk;
    if (lane == 0) { iters_out[w] = iters; change_out[w] = max_change; }

    for (int s = 0; s < steps; ++s) {                                        // :773-797
        const int i = s * G + slot;
        const bool mine = live && i < W.n_ind;
        const double p = s == 0 ? P0 : element(s);
        const int has = s == 0 ? D0 : with_data(s);
        double y;
        const double sum = matvec(p, fr_prev, fk_prev, y);
        double bv = 0.0;
        int bj = 0x7fffffff;
        if (mine && r <= k) {                                                // genotype (a, b) = (r, k)
            double v = p * fr_prev * fk_prev * (1 + (r != k));               // :847
            if (sum > 0.0) v /= sum;
            bv = v;
            bj = (int)(tri_row_start(r, n) + k - r);
        }
        double best_v = 0.0;                                                 // largest value of the slot, the first
        int best_j = 0x7fffffff;                                             // genotype among equal ones
        for (int jj = 0; jj < nn; ++jj) {
            const double ov = __shfl(bv, sb + jj);
            const int oj = __shfl(bj, sb + jj);
            if (oj != 0x7fffffff && (best_j == 0x7fffffff || ov > best_v || (ov == best_v && oj < best_j))) { best_v = ov; best_j = oj; }
        }
        if (mine && e == 0) {
            if (!has) { best_gen[W.ind_base + i] = -1; best_val[W.ind_base + i] = 0.0; }
            else if (iters == 0) { best_gen[W.ind_base + i] = 0; best_val[W.ind_base + i] = 0.0; }
            else { best_gen[W.ind_base + i] = best_j; best_val[W.ind_base + i] = best_v; }
        }
    }
}

__global__ __launch_bounds__(256) void k_em_small(const WinDesc* __restrict__ wins, int n_win,
                                                  const int32_t* __restrict__ ioff,
                                                  const double* __restrict__ full,
                                                  int max_iters,
                                                  double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                                  double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                                  double* __restrict__ best_val)
{
    const int w = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
    if (w >= n_win) return;
    const int lane = threadIdx.x & 63;
    const WinDesc W = wins[w];
    if (W.n_ind > 1 && W.n_hap * W.n_hap <= 32)
        em_small_slots(W, w, lane, ioff, full, max_iters, freq_out, iters_out, change_out, best_gen, best_val);
    else
        em_small_one(W, w, lane, ioff, full, max_iters, freq_out, iters_out, change_out, best_gen, best_val);
}

// ---------------------------------------------------------------------------
// k_em_exact: the windows em_is_exact names (at most 8 haplotypes, or 16 while small) with every sum of platypus/cpopulation.pyx:807-890 in the
// REFERENCE'S OWN ORDER, so that the frequencies are the doubles a sequential run of that text gives for the same
// genotype posteriors (the kernels above regroup the sums into matrix-vector products and agree to 1e-9; a
// frequency that a sequential run makes exactly 0.5 came out one bit off there, and a p-value printed from it
// could turn on that bit).  One wavefront per window.  Per iteration: every (individual, genotype) value
// P f_a f_b (1 + (a != b)) by the lanes side by side (:847); every individual's csrSum by one lane, genotypes in
// order (:849); the divisions side by side (:852-854); then lane k adds up haplotype k's total over individuals
// and genotypes in order (:858-880: each total receives the same terms in the same order as the reference's loop
// nest, which is all its value depends on).  The values live in LDS: n_ind * (n_gen + 1) + kExactPad doubles per
// wavefront (`slice` doubles apart).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A window with ONE individual (the usual case of single-sample calling) needs no LDS: lane j holds genotype j's
// value, csrSum is added up through lane reads in genotype order, and lane k collects haplotype k's n terms, again
// in genotype order, through lane permutes (every lane runs the same n steps: a permute reads live lanes only).
__device__ __forceinline__ void em_exact_one(const WinDesc& W, int w, int lane, const int32_t* __restrict__ io,
                                             const double* __restrict__ Pw, int max_iters,
                                             double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                             double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                             double* __restrict__ best_val)
{
    const int n = W.n_hap, n_gen = W.n_gen;
    const bool has = io[1] != io[0];
    const bool gen = lane < n_gen;
    const int a = gen ? tri_row_of(lane, n) : 0;
    const int b = gen ? a + (lane - (int)tri_row_start(a, n)) : 0;
    const double p = gen ? Pw[a * W.ld + b] : 0.0;
    const double twice = (double)(1 + (a != b));
    const int k = lane < n ? lane : 0;
    const int kk = (int)tri_row_start(k, n);
    double eps = 1.0 / (1 * 2 * 2);                                          // :751
    eps = eps < 1e-3 ? eps : 1e-3;
    double max_change = eps + 1;
    int iters = 0;
    double fk = 1.0 / n;                                                     // lanes 0..n-1: f_k (every lane starts with the same value)
    double c = 0.0;
    while (max_change > eps && iters < max_iters) {                          // :766-768
        double nf = 0.0;                                                     // no reads: the sums stay empty and 0 / 0 ... see below
        if (has) {
            const double fa = __shfl(fk, a), fb = __shfl(fk, b);
            const double v = gen ? __dmul_rn(__dmul_rn(__dmul_rn(p, fa), fb), twice) : 0.0;      // :847
            double sum = 0.0;
            for (int j = 0; j < n_gen; ++j) sum = __dadd_rn(sum, __shfl(v, j));                  // :849
            c = sum > 0.0 ? __ddiv_rn(v, sum) : v;                                               // :852-854
            double t = 0.0;
            for (int m = 0; m < n; ++m) {                                    // :858-880: haplotype k's terms in genotype order
                const int j = m < k ? (int)tri_row_start(m, n) + k - m : kk + m - k;
                const double term = __shfl(c, j);
                t = __dadd_rn(t, m == k ? __dmul_rn(2.0, term) : term);
            }
            nf = __ddiv_rn(t, 2.0);                                          // :882-884, one individual with data
        } else {
            nf = __ddiv_rn(0.0, 0.0);                                        // tmp / (ploidy * 0): the reference divides by zero here
        }
        const double ch = lane < n ? fabs(fk - nf) : 0.0;
        max_change = wave_max(ch);                                           // :885-888
        fk = nf;
        ++iters;
    }
    if (lane < n) freq_out[W.hap_base + lane] = fk;
    if (lane == 0) { iters_out[w] = iters; change_out[w] = max_change; }
    int best = -1;
    double mx = 0.0;
    for (int j = 0; j < n_gen; ++j) {                                        // :773-797
        const double v = __shfl(c, j);
        if (best == -1 || v > mx) { mx = v; best = j; }
    }
    if (lane == 0) {
        if (!has) { best_gen[W.ind_base] = -1; best_val[W.ind_base] = 0.0; }
        else if (iters == 0) { best_gen[W.ind_base] = 0; best_val[W.ind_base] = 0.0; }
        else { best_gen[W.ind_base] = best; best_val[W.ind_base] = mx; }
    }
}

__global__ __launch_bounds__(256) void k_em_exact(const WinDesc* __restrict__ wins, int n_win,
                                                  const int32_t* __restrict__ ioff,
                                                  const double* __restrict__ full,
                                                  int max_iters, int slice,
                                                  double* __restrict__ freq_out, int32_t* __restrict__ iters_out,
                                                  double* __restrict__ change_out, int32_t* __restrict__ best_gen,
                                                  double* __restrict__ best_val)
{
    extern __shared__ double em_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wave));
    if (w >= n_win) return;
    const WinDesc W = wins[w];
    if (!em_is_exact(W.n_hap, W.n_ind, W.n_gen)) return;                     // a wider window of a mixed batch: the other kernels' work
    const int n = W.n_hap, n_gen = W.n_gen, n_ind = W.n_ind, cells = n_ind * n_gen;
    const int32_t* io = ioff + W.ioff_base;
    const double* Pw = full + W.full_base;
    if (n_ind == 1 && n_gen <= 64) {
        em_exact_one(W, w, lane, io, Pw, max_iters, freq_out, iters_out, change_out, best_gen, best_val);
        return;
    }
    double* c = em_lds + (size_t)wave * slice;                                // csr[individual][genotype]
    double* cs = c + cells;                                                  // csrSum per individual
    double* f = cs + n_ind;                                                  // frequencies [16]
    int* gtab = reinterpret_cast<int*>(f + kExactMaxHapsLight);              // genotype -> a | b << 8   (up to 136)
    for (int j = lane; j < n_gen; j += 64) {
        const int a = tri_row_of(j, n);
        gtab[j] = a | ((a + (j - (int)tri_row_start(a, n))) << 8);
    }
    if (lane < n) f[lane] = 1.0 / n;                                         // :753,763-764
    int n_with_data = 0;
    for (int i = 0; i < n_ind; ++i) n_with_data += io[i + 1] != io[i];
    double eps = 1.0 / (n_ind * 2 * 2);                                      // :751
    eps = eps < 1e-3 ? eps : 1e-3;
    double max_change = eps + 1;
    int iters = 0;
    wave_sync();
    while (max_change > eps && iters < max_iters) {                          // :766-768
        for (int idx = lane; idx < cells; idx += 64) {
            const int i = idx / n_gen, j = idx - i * n_gen;
            if (io[i + 1] == io[i]) continue;                                // :834-835
            const int g = gtab[j], a = g & 0xff, b = g >> 8;
            const double p = Pw[(int64_t)i * n * W.ld + a * W.ld + b];
            c[idx] = __dmul_rn(__dmul_rn(__dmul_rn(p, f[a]), f[b]), (double)(1 + (a != b)));   // :847, left to right
        }
        wave_sync();
        for (int i = lane; i < n_ind; i += 64) {
            if (io[i + 1] == io[i]) continue;
            double sum = 0.0;
            for (int j = 0; j < n_gen; ++j) sum = __dadd_rn(sum, c[i * n_gen + j]);   // :849
            cs[i] = sum;
        }
        wave_sync();
        for (int idx = lane; idx < cells; idx += 64) {
            const int i = idx / n_gen;
            if (io[i + 1] == io[i]) continue;
            const double sum = cs[i];
            if (sum > 0.0) c[idx] = __ddiv_rn(c[idx], sum);                  // :852-854
        }
        wave_sync();
        double nf = 0.0, ch = 0.0;
        if (lane < n) {
            const int k = lane;
            const int kk = (int)tri_row_start(k, n);
            double t = 0.0;
            for (int i = 0; i < n_ind; ++i) {                                // :858-880, the terms of haplotype k in order
                if (io[i + 1] == io[i]) continue;
                const double* ci = c + i * n_gen;
                for (int a = 0; a < k; ++a) t = __dadd_rn(t, ci[(int)tri_row_start(a, n) + k - a]);
                t = __dadd_rn(t, __dmul_rn(2.0, ci[kk]));
                for (int b = k + 1; b < n; ++b) t = __dadd_rn(t, ci[kk + b - k]);
            }
            nf = __ddiv_rn(t, (double)(2 * n_with_data));                    // :882-884
            ch = fabs(f[k] - nf);
        }
        max_change = wave_max(ch);                                           // :885-888
        wave_sync();
        if (lane < n) f[lane] = nf;
        wave_sync();
        ++iters;
    }
    if (lane < n) freq_out[W.hap_base + lane] = f[lane];
    if (lane == 0) { iters_out[w] = iters; change_out[w] = max_change; }
    for (int i = lane; i < n_ind; i += 64) {                                 // :773-797: the first largest csr of the last iteration
        if (io[i + 1] == io[i]) { best_gen[W.ind_base + i] = -1; best_val[W.ind_base + i] = 0.0; continue; }
        if (iters == 0) { best_gen[W.ind_base + i] = 0; best_val[W.ind_base + i] = 0.0; continue; }
        int best = -1;
        double mx = 0.0;
        for (int j = 0; j < n_gen; ++j) {
            const double v = c[i * n_gen + j];
            if (best == -1 || v > mx) { mx = v; best = j; }
        }
        best_gen[W.ind_base + i] = best; best_val[W.ind_base + i] = mx;
    }
}

// ---------------------------------------------------------------------------
// k_var_post: Population.calculatePosterior, platypus/cpopulation.pyx:515-632.
// One wavefront per variant.  has_var marks the haplotypes that carry the
// variant; f' = f restricted to the others and rescaled (:554-580).  Per
// individual the two sums over genotypes (:597-614) are f.(P f) and f'.(P f').
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_var_post(const WinDesc* __restrict__ wins,
                                                 const int32_t* __restrict__ ioff,
                                                 const double* __restrict__ full,
                                                 const double* __restrict__ freq,
                                                 const int32_t* __restrict__ var_win,
                                                 const int64_t* __restrict__ var_mask_off,
                                                 const uint8_t* __restrict__ has_var,
                                                 const double* __restrict__ prior,
                                                 double* __restrict__ out, int n_max)
{
    extern __shared__ double vp_lds[];
    double* f = vp_lds;
    double* fp = f + n_max;
    const int v = blockIdx.x, lane = threadIdx.x;
    const WinDesc W = wins[var_win[v]];
    const int n = W.n_hap;
    const uint8_t* m = has_var + var_mask_off[v];
    const int32_t* io = ioff + W.ioff_base;
    double part = 0.0;
    for (int k = lane; k < n; k += 64) {
        const double fk = freq[W.hap_base + k];
        f[k] = fk;
        fp[k] = m[k] ? 0.0 : fk;
        part += fp[k];
    }
    // The windows em_is_exact names get every sum in the reference's own order (:554-627), one individual per lane:
    // variants whose inputs are equal then have exactly equal posteriors, and the ranking by posterior
    // (variantcaller.pyx:157-165) breaks their ties the way the reference does.
    const bool exact = em_is_exact(n, W.n_ind, W.n_gen);
    double sum_freqs = wave_sum(part);
    __builtin_amdgcn_wave_barrier();
    if (exact) {
        sum_freqs = 0.0;
        for (int k = 0; k < n; ++k)
            if (!m[k]) sum_freqs = __dadd_rn(sum_freqs, f[k]);
    }
    if (sum_freqs > 0)
        for (int k = lane; k < n; k += 64)
            if (fp[k] != 0.0) fp[k] /= sum_freqs;
    __builtin_amdgcn_wave_barrier();
    double slp_var = 0.0, slp_novar = 0.0;
    if (exact) {
        for (int i0 = 0; i0 < W.n_ind; i0 += 64) {
            const int i = i0 + lane;
            double lv = 0.0, ln = 0.0;
            int has = 0;
            if (i < W.n_ind && io[i + 1] != io[i]) {                         // :590-591
                has = 1;
                const double* P = full + W.full_base + (int64_t)i * n * W.ld;
                double pv = 0.0, pn = 0.0;
                for (int r = 0; r < n; ++r)
                    for (int q = r; q < n; ++q) {                            // :597-614, genotype by genotype
                        const double g = P[(int64_t)r * W.ld + q];
                        const double fac = r != q ? 2.0 : 1.0;
                        pv = __dadd_rn(pv, __dmul_rn(__dmul_rn(__dmul_rn(fac, f[r]), f[q]), g));
                        pn = __dadd_rn(pn, __dmul_rn(__dmul_rn(__dmul_rn(fac, fp[r]), fp[q]), g));
                    }
                lv = pv > 0 ? log(pv) : -708.0;                              // :616-624
                ln = pn > 0 ? log(pn) : -708.0;
            }
            const int here = min(64, W.n_ind - i0);
            for (int j = 0; j < here; ++j) {                                 // added up in the individuals' order
                const int hj = __shfl(has, j);
                const double a = __shfl(lv, j), b = __shfl(ln, j);
                if (hj) { slp_var = __dadd_rn(slp_var, a); slp_novar = __dadd_rn(slp_novar, b); }
            }
        }
        if (lane == 0) {
            const double ratio = exp(slp_novar - slp_var);                   // :626-627
            const double pr = prior[v];
            out[v] = pr / (pr + ratio * (1.0 - pr));
        }
        return;
    }
    for (int i = 0; i < W.n_ind; ++i) {
        if (io[i + 1] == io[i]) continue;                                    // :590-591
        const double* P = full + W.full_base + (int64_t)i * n * W.ld;
        double pv = 0.0, pn = 0.0;
        for (int k = lane; k < n; k += 64) {
            double y = 0.0, yp = 0.0;
            for (int r = 0; r < n; ++r) {
                const double p = P[(int64_t)r * W.ld + k];
                y += p * f[r];
                yp += p * fp[r];
            }
            pv += f[k] * y;
            pn += fp[k] * yp;
        }
        pv = wave_sum(pv);
        pn = wave_sum(pn);
        slp_var += pv > 0 ? log(pv) : -708.0;                                // :616-624
        slp_novar += pn > 0 ? log(pn) : -708.0;
    }
    if (lane == 0) {
        const double ratio = exp(slp_novar - slp_var);                       // :626-627
        const double pr = prior[v];
        out[v] = pr / (pr + ratio * (1.0 - pr));
    }
}

}  // namespace pg
