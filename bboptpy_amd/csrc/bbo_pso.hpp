// bbo_pso.hpp -- device-resident adaptive PSO (Zhan et al. 2009).
//
// Reference: APSOSearch (src/multivariate/pso/apso.cpp:48-452).  The reference moves one
// particle at a time and refreshes the global best inside that loop; here a generation is
//   pso_center + pso_ese   evolutionary factor f (apso.cpp:300-339): all-pairs mean distance
//                          as a Gram matrix of the CENTRED swarm on the fp64 matrix cores
//   pso_control_a/b        fuzzy state machine, w/c1/c2, elitist learning (:200-298,:347-452)
//   pso_update             fused velocity / position / clamp / evaluate / pbest (:159-198)
//   pso_finish             gbest arg-min, stop test (:129-145)
// with the generation-synchronous semantics stated in DESIGN.md and restated on the CPU by
// oracle/bbo_oracle_pop.inc (Apso with sync = true).
#pragma once

#include "bbo_common.hpp"

namespace bbo {

struct PsoScal {
    double w, c1, c2;        // inertia / acceleration (apso.cpp:64-66)
    double fbest;            // fitness of xbest
    double evof;             // evolutionary factor of the last generation
    double nu;               // fitness of the elitist candidate
    double m2;               // radius spread of the last stop test
    int state;               // 0 at start, then 1..4 (0 again after state 4: see apso.cpp:384)
    int it, fev, maxit;
    int stop, conv;
    int need_elite;          // this generation runs elitist learning
    int ibest_cur;           // particle with the best CURRENT fitness (getf's ibest)
    int bad_rule;            // the reference would throw std::invalid_argument (apso.cpp:381)
    int pad_;
};

struct PsoConst {
    int n, ld, np, correct, obj, mfev, honor_stop, npop;
    int ldc, npad;           // Xc: row stride (n rounded up to 16), rows per population (np to 128)
    double tol;
    uint64_t seed;
};

struct PsoDev {
    double *X, *V, *XB;      // [P][np][ld]
    double *f, *fb;          // [P][np]
    double *xbest;           // [P][ld]
    double *ws;              // [P][np] mean distance to the others
    double *mean;            // [P][ld] swarm centroid
    double *nrm;             // [P][np] squared norm of the centred particle
    double *Xc;              // [P][npad][ldc] centred swarm, zero padded (operands of pso_ese_sym)
    double *pvec;            // [P][ld] elitist candidate
    double *radius;          // [P][np]
    double *colpart;         // [P][parts][ld] centroid partial sums
    double *colpart2;        // [P][ceil(np/128)][np] distance column sums of block I (pso_ese_sym)
    double *rowpart2;        // [P][np] distance row sums inside the swept blocks
    const double *lower, *upper, *aux;
    PsoScal *scal;
};

class PsoEngine: public Optimizer {
public:
    explicit PsoEngine(const bbo_params &p);
    ~PsoEngine() override;
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
    void host_evaluate_swarm(int i0 = 0, int i1 = -1);
    void host_evaluate_elite();
    bool all_stopped();

    bbo_params params_;
    ObjectiveSpec obj_;
    PsoConst c_ {};
    PsoDev d_ {};
    hipStream_t stream_ = nullptr;
    bool inited_ = false;
    int parts_ = 1;
    int chunk_ = 0;           // particles between two refreshes of the swarm's best inside a generation
    std::vector<double> aux_h_;
    DevBuf<double> colpart2_, rowpart2_, Xc_;
    DevBuf<double> X_, V_, XB_, f_, fb_, xbest_, ws_, mean_, nrm_, pvec_, radius_, colpart_,
            lower_, upper_, aux_;
    DevBuf<PsoScal> scal_;
    KernelTimer timer_;
};

} // namespace bbo
