// bbo_cma.hpp -- device-resident CMA-ES / active CMA-ES engine (declarations).
//
// One generation of the reference (BaseCmaes::iterate, base_cmaes.cpp:150-156:
// samplePopulation -> evaluateAndSortPopulation -> updateDistribution ->
// updateHistory) is a fixed sequence of kernels over state that never leaves HBM.
// See DESIGN.md for the kernel table and the data layout.
#pragma once

#include "bbo_common.hpp"

namespace bbo {

// per-population scalars, updated by the kernels only
struct CmaScal {
    double sigma;
    double fbest, fworst;     // running best/worst over the history ring (base_cmaes.cpp:201-208)
    double pslen;             // ||ps|| of the last update (diagnostic)
    double ybw[4];            // best, 2nd best, 2nd worst, worst fitness (base_cmaes.cpp:226-229)
    int ibw[4];
    int it, fev;
    int flag;                 // reference stop flag 1..9 (cmaes.cpp:151-227), 0 = none
    int stop;                 // sticky: 1 = stop rule fired, 2 = evaluation budget exhausted
    int hsig;
    int eigenlastev, eigen_done;
    int hist_head, hist_len;
    int basis_ok;             // C^-1/2 = B D^-1 B^T for the (B, D) the sampler uses (see cma_whiten128)
    int eig_stage;            // 128 < n <= 256, split decomposition: 1 = tridiagonal form handed to the next kernels
    int eig_mw_fail;          // sticky: a wavefront of cma_tred_mw gave up waiting for its partners (bbo_eig_mw.hpp)
};

// strategy constants, passed to every kernel by value
struct CmaConst {
    int n, ld;                // dimension, padded leading dimension (multiple of 16)
    int lambda, lambda_pad;   // population size, padded to a multiple of 16
    int mu, mu_pad;
    int variant;              // 0 plain (cmaes.cpp), 1 active (active_cmaes.cpp), 2 separable (sep_cmaes.cpp)
    int bound, obj;
    int use_zn;               // this generation's zn2 is valid and x was not clamped
    int lazy_isc;             // C^-1/2 is not formed after a decomposition (16 < ld <= 256, no box): cma_paths
                              // works from B and D, the whitening takes ||z||^2, readers get it on demand
    int mfev, mit, hlen, ik;
    int honor_stop;           // 1 inside run()/optimize(): stopped populations are frozen
    int splits, rps;          // Gram split-K: number of row slabs, rows per slab
    int npop;
    double mueff, cc, cs, c1, cmu, cneg, alphaold, cm, damps, chi, sigma0, tol, eigenfreq;
    double ccov;              // separable variant (sep_cmaes.cpp:53-62)
    uint64_t seed;
    // extensions, off by default (bbo_set "stop_off" / "ftarget"): bit k of stop_off silences the
    // reference's stop test with flag k; f_best <= ftarget raises the non-reference flag 10
    int stop_off;
    double ftarget;
};

struct CmaDev {
    double *X;          // [P][lambda_pad][ld]   candidates (arx)
    double *f;          // [P][lambda_pad]       fitness, +inf on padding rows
    int *rank;          // [P][lambda_pad]       rank of candidate i
    int *order;         // [P][lambda_pad]       candidate with rank r   (= _fitness[r]._index)
    double *xmean, *xold, *pc, *ps;   // [P][ld]
    double *C;          // [P][ld][ld]  covariance, symmetric full (reference keeps the lower half)
    double *B;          // [P][ld][ld]  eigenvectors in columns
    double *D;          // [P][ld]      sqrt(eigenvalues), ascending
    double *isc;        // [P][ld][ld]  C^-1/2
    double *BDp;        // [P][ld*ld]   (B diag D) in MFMA B-fragment order
    double *ISp;        // [P][ld*ld]   C^-1/2   in MFMA B-fragment order
    double *S;          // [P][mu_pad]  whitened squared norms of the worst mu
    double *csep;       // [P][ld] diagonal covariance of the separable variant (D = its sqrt)
    double *zn2;        // [P][lambda_pad] ||z||^2 of every candidate (cma_sample_eval128 only)
    double *gram_part;  // [P][splits][ld][ld]
    double *mean_part;  // [P][splits][ld]
    double *hist_best, *hist_kth;     // [P][hlen]
    double *eig_work;   // [P][ld][ldw] scratch for the eigensolver when it does not fit LDS
    const double *weights;            // [mu]
    const double *lower, *upper, *aux;   // [ld]
    const double *zinject;            // [P][lambda][n] or null
    double *zrecord;                  // [P][lambda][n] or null
    CmaScal *scal;                    // [P]
    long long *stamps;                // [16] eigensolver phase clocks (diagnostic) or null
    int dbg;                          // diagnostic switches (0 in production)
    int mw_fault;                     // fault injection of bbo_eig_mw.hpp (-1: none; environment only)
    int *mw_fail_host;                // pinned host word: a spread reduction of this engine timed out
};

class CmaEngine: public Optimizer {
public:
    explicit CmaEngine(const bbo_params &p);
    ~CmaEngine() override;

    void init(int n, const double *lower, const double *upper, const double *guess,
            const ObjectiveSpec &obj) override;
    void iterate() override;
    void solution(int population, double *x_out, int *n_evals, int *converged) override;
    void optimize(int n, const double *lower, const double *upper, const double *guess,
            const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged) override;
    int run(int max_generations) override;
    int get(const std::string &key, int population, double *out, int cap) override;
    int set(const std::string &key, int population, const double *in, int count) override;
    int dimension() const override { return c_.n; }

    // BaseCmaes::setParams (base_cmaes.cpp:136-148), used by the restart drivers
    void set_params(int np, double sigma, int mfev);
    void phase(int which);
    void inject_normals(const double *z, int count);
    // evaluates one point with this engine's objective (restart drivers' extra call)
    double evaluate_point(const double *x);
    int lambda() const { return params_.np; }
    int populations() const { return params_.populations; }
    uint64_t seed() const { return params_.seed; }
    void set_seed(uint64_t s) { params_.seed = s; }
    // the next init() starts like a new object: B = C = I (not the previous run's, cmaes.cpp:53-59)
    void fresh_start(uint64_t s)
    {
        params_.seed = s;
        keep_bc_ = false;
    }

private:
    void generation(bool honor_stop);
    bool small_fused_ok() const;
    void launch_small(int gens, bool honor_stop);
    void launch_sample_eval();
    void launch_post(int mode);
    void launch_rank();
    void launch_update();
    void launch_eigen();
    void launch_history_stop();
    void host_evaluate();
    void fetch_scal(std::vector<CmaScal> &out);
    bool all_stopped();

    bbo_params params_;
    ObjectiveSpec obj_;
    CmaConst c_ {};
    CmaDev d_ {};
    hipStream_t stream_ = nullptr;
    bool inited_ = false;
    bool keep_bc_ = false;    // B and C survive a re-init of the same object (cmaes.cpp:53-54)
    bool basis_maybe_stale_ = false;   // some population's basis_ok may be 0 (refreshed at every poll)
    // the Householder reduction spread over several workgroups (128 < n <= 256, few populations):
    // its exchange buffers, the epoch base of its flags, and 'a wavefront once gave up: do not use it'
    DevBuf<double> mw_buf_;
    unsigned long long mw_launch_ = 0;
    bool mw_disabled_ = false;
    int mw_xcd_ = next_mw_xcd();        // this engine's offset into the XCDs
    static int next_mw_xcd();
    long sample_wide_max_tiles_ = 512;       // n = 128: at most this many 16-row tiles take cma_sample_eval<1, 8>
    long sample128_min_rows_ = 256 * 128;   // candidates in flight from which cma_sample_eval128 is used
    int split_maxp_ = 32;              // 64 < n <= 128: at most this many populations take the split decomposition
    bool rank_wrote_norms_ = false;    // this generation's cma_rank_sort wrote S: no whiten launch
    int last_n_ = -1;
    std::vector<double> lower_h_, upper_h_, aux_h_;

    DevBuf<double> zn2_, csep_;
    DevBuf<double> X_, f_, xmean_, xold_, pc_, ps_, C_, B_, D_, isc_, BDp_, ISp_, S_,
            gram_part_, mean_part_, hist_best_, hist_kth_, eig_work_, weights_, lower_,
            upper_, aux_, zinject_, zrecord_;
    DevBuf<int> rank_, order_;
    DevBuf<long long> stamps_;
    DevBuf<CmaScal> scal_;
    int *mw_fail_host_ = nullptr;   // pinned, device-visible: raised by a spread reduction that timed out
    long mw_reserved_ = 0;          // workgroups this engine holds of the device's MwBudget
    bool mw_launched_ = false;      // a spread kernel went out since the flag was last read
    bool mw_reserve(long workgroups);
    void mw_release();
    bool mw_check_failed();         // after a synchronisation: true once when the flag went up
    KernelTimer timer_;
};

} // namespace bbo
