// bbo_de.hpp -- device-resident differential evolution: L-SHADE, JADE and SaNSDE.
//
// Reference: ShadeSearch (src/multivariate/de/shade.cpp:56-298) and JadeSearch
// (src/multivariate/de/jade.cpp:64-294).  The reference loops over the individuals one at a
// time and replaces them in place (asynchronous); here one generation is ONE fused,
// HBM-bound kernel over the whole population (draw parameters -> mutate/crossover ->
// bound repair -> evaluate -> select into the other half of a double buffer), followed by
// small bookkeeping kernels.  The generation-synchronous semantics are stated in DESIGN.md
// and restated on the CPU by oracle/bbo_oracle_pop.inc (De::iterate_sync).
#pragma once

#include "bbo_common.hpp"

namespace bbo {

struct DeScal {
    double mucr, muf;        // JADE adaptive means (jade.cpp:78-79)
    double m2;               // sum of squared radius deviations of the last stop test
    int np;                  // current population size (L-SHADE shrinks it, shade.cpp:218-225)
    int larch;               // archive fill
    int k;                   // 1-based memory slot (shade.cpp:207-211)
    int fev, gen;
    int stop;                // sticky: 1 = radius test fired, 2 = evaluation budget exhausted
    int conv;                // result of the last stop test
    int cur;                 // which half of the double buffer holds the population
    int nsucc;               // successes of the last generation
    int pad_;
    // SaNSDE adaptation state (sansde.cpp:70-80): strategy / mutation-kind probabilities, CR mean
    double sp, sfp, crm, crrec, crdeltaf;
    double fpns[2], fpnf[2];
    int pns[2], pnf[2];
};

struct DeConst {
    int variant;             // 0 = L-SHADE, 1 = JADE, 2 = SaNSDE
    int ncrref, npup, ncrup; // SaNSDE: generations between CR refresh / p update / CR-F update
    int n, ld;
    int npinit, npmin, h;
    int archive, repaircr;
    int obj, mfev, honor_stop, npop;
    int np_launch;           // grid size in individuals (host-tracked upper bound of np)
    double tol, pelite, cdamp, jsigma;
    uint64_t seed;
};

struct DeDev {
    double *X[2];            // [P][npinit][ld] double-buffered population (sorted position -> row)
    double *f[2];            // [P][npinit]
    int *order;              // [P][npinit] sorted position -> physical row of X[cur]
    int *rank;               // [P][npinit]
    double *arch;            // [P][npinit][ld]
    double *MCR, *MF;        // [P][h]
    double *rec_cr, *rec_f, *rec_df, *radius;   // [P][npinit]
    int *rec_flag;           // [P][npinit] bit0 accepted, bit1 strictly better
    int *claim;              // [P][npinit] archive slot -> highest claiming individual (or -1)
    int *slot_of;            // [P][npinit] individual -> archive slot it writes (or -1);
                             // SaNSDE: bit 0 = mutation strategy, bit 1 = F distribution of the trial
    double *crow[2];         // SaNSDE [P][npinit] per-individual CR, double buffered like X
    const double *lower, *upper, *aux;
    DeScal *scal;
};

class DeEngine: public Optimizer {
public:
    explicit DeEngine(const bbo_params &p);
    ~DeEngine() override;
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

private:
    void generation(bool honor_stop);
    void launch_rank(int which_next, int np_bound);
    // threads of the one-workgroup-per-population kernels (bookkeeping, finish): one per
    // individual up to 1024 -- sums come out bit-identical whatever the choice as long as every
    // thread holds at most one individual, and small populations do not pay for 16 wavefronts
    int pop_threads() const { return c_.npinit <= 64 ? 64 : c_.npinit <= 256 ? 256 : 1024; }
    void host_evaluate(int which, int rows);
    bool all_stopped();

    bbo_params params_;
    ObjectiveSpec obj_;
    DeConst c_ {};
    DeDev d_ {};
    hipStream_t stream_ = nullptr;
    bool inited_ = false;
    int np_host_ = 0;        // upper bound of the device np (exact while no population stopped)
    long fev_host_ = 0;
    std::vector<double> aux_h_;
    DevBuf<double> cra_, crb_;
    DevBuf<double> Xa_, Xb_, fa_, fb_, arch_, MCR_, MF_, rec_cr_, rec_f_, rec_df_, radius_,
            lower_, upper_, aux_;
    DevBuf<int> order_, rank_, rec_flag_, claim_, slot_of_;
    DevBuf<DeScal> scal_;
    KernelTimer timer_;
};

} // namespace bbo
