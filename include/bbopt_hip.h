/*
 * bbopt_hip.h -- C ABI of libbbopt_hip.so, the MI355X (gfx950) population-based
 * black-box optimizer core.
 *
 * The reference (mike-gimelfarb/bboptpy) has NO C ABI: its plugin surface is the
 * C++ abstract class MultivariateOptimizer bound to Python with pybind11
 *     /root/reference/src/multivariate/multivariate.h:132-146   (init / iterate /
 *                                                                solution / optimize)
 *     /root/reference/py/multivariate_py.cpp:374-420            (MultivariateSearch)
 * This header is the C statement of exactly that four-method interface plus the
 * constructor argument lists of the algorithms on the hot path.  Every entry
 * point names the reference interface it replaces.  Plain pointers and sizes
 * only; all device memory, HIP streams and kernels live behind the handle.
 *
 * Threading: one host thread per handle at a time (the reference is single
 * threaded and holds the GIL for a whole optimize(), multivariate_py.cpp:376-395).
 * Errors: every function returns BBO_OK (0) or a negative bbo_status;
 * bbo_last_error() gives the text (the reference throws std::invalid_argument /
 * propagates Python exceptions instead, apso.cpp:381,448).
 */
#ifndef BBOPT_HIP_H_
#define BBOPT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbo_handle_s *bbo_handle;

typedef enum {
    BBO_OK = 0,
    BBO_ERR_ARG = -1,        /* bad argument / unsupported shape                 */
    BBO_ERR_STATE = -2,      /* call out of order (e.g. iterate before init)     */
    BBO_ERR_HIP = -3,        /* a HIP runtime call or kernel failed              */
    BBO_ERR_NO_DEVICE = -4,  /* no gfx950 device visible: there is NO CPU path   */
    BBO_ERR_CALLBACK = -5,   /* the host objective callback reported failure     */
    BBO_ERR_KEY = -6         /* unknown state key in bbo_get / bbo_set           */
} bbo_status;

/* Algorithms on the hot path; names are the reference's Python class names
 * (py/multivariate_py.cpp:103-115,137-171,265-269). */
typedef enum {
    BBO_ALGO_CMAES = 0,        /* Cmaes        src/multivariate/cma/cmaes.h:40        */
    BBO_ALGO_ACTIVE_CMAES = 1, /* ActiveCmaes  src/multivariate/cma/active_cmaes.h:40 */
    BBO_ALGO_SHADE = 2,        /* ShadeSearch  src/multivariate/de/shade.h:42         */
    BBO_ALGO_JADE = 3,         /* JadeSearch   src/multivariate/de/jade.h:49          */
    BBO_ALGO_APSO = 4,         /* APSOSearch   src/multivariate/pso/apso.h:38         */
    BBO_ALGO_IPOP_CMAES = 5,   /* IPopCmaes    src/multivariate/cma/ipop_cmaes.h:55   */
    BBO_ALGO_BIPOP_CMAES = 6,  /* BiPopCmaes   src/multivariate/cma/bipop_cmaes.h:49  */
    BBO_ALGO_SEP_CMAES = 7,    /* SepCmaes     src/multivariate/cma/sep_cmaes.h:36    */
    BBO_ALGO_SANSDE = 8,       /* SaNSDESearch src/multivariate/de/sansde.h:40        */
    BBO_ALGO_CSO = 9,          /* CSOSearch    src/multivariate/pso/cso.h:44          */
    BBO_ALGO_CCPSO = 10        /* CCPSOSearch  src/multivariate/pso/ccpso.h:46        */
} bbo_algo;

/* Built-in objectives evaluated on the device (the reference ships none; id 1 is
 * the README objective, README.md:111-112).  Same ids as oracle/objectives.h. */
typedef enum {
    BBO_OBJ_SPHERE = 0,
    BBO_OBJ_ROSENBROCK = 1,
    BBO_OBJ_RASTRIGIN = 2,
    BBO_OBJ_ELLIPSOID = 3,
    BBO_OBJ_ACKLEY = 4,
    BBO_OBJ_GRIEWANK = 5,
    BBO_OBJ_CIGAR = 6,
    BBO_OBJ_DISCUS = 7,
    BBO_OBJ_DIFFPOW = 8,
    BBO_OBJ_SCHWEFEL12 = 9
} bbo_objective_id;

/* Host objective callbacks -- the compatibility path for an arbitrary Python `f`
 * (replaces the std::function<double(const double*)> of multivariate.h:32 and the
 * per-evaluation NumPy wrapper of multivariate_py.cpp:385-388).
 *   scalar: one candidate, returns f(x); set *failed != 0 to abort the run.
 *   batch:  `rows` candidates of `n` coordinates, row stride `ld` doubles;
 *           writes f_out[rows]; returns 0 on success.                          */
typedef double (*bbo_scalar_fn)(const double *x, int n, void *user, int *failed);
typedef int (*bbo_batch_fn)(const double *X, int rows, int n, int ld, double *f_out,
        void *user);

typedef enum {
    BBO_OBJECTIVE_BUILTIN = 0,
    BBO_OBJECTIVE_SCALAR_CALLBACK = 1,
    BBO_OBJECTIVE_BATCH_CALLBACK = 2
} bbo_objective_kind;

typedef struct {
    int kind;              /* bbo_objective_kind                                  */
    int builtin;           /* bbo_objective_id when kind == BUILTIN               */
    bbo_scalar_fn scalar;
    bbo_batch_fn batch;
    void *user;
} bbo_objective;

/* Constructor arguments.  Field names and defaults are the reference's keyword
 * arguments (py/multivariate_py.cpp): zero-initialise, call bbo_params_default(),
 * then overwrite.  Fields an algorithm does not use are ignored. */
typedef struct {
    int algo;              /* bbo_algo                                            */
    /* shared */
    int mfev;              /* maximum objective evaluations                       */
    double tol;            /* stop tolerance (meaning differs per family, SURVEY A-17) */
    int np;                /* population size: CMA `np` (lambda), JADE/APSO `np`, SHADE `npinit` */
    /* CMAES(mfev,tol,np,sigma0=2,bound=False,eigenrate=0.25)            :103-108
     * ActiveCMAES(...,alphacov=2,eigenrate=0.25)                         :110-115
     * SepCMAES(mfev,tol,np,sigma0=2,bound=False,adjustlr=True)          :131-135 */
    double sigma0;
    int bound;
    double alphacov;
    double eigenrate;
    /* JADE(mfev,np,tol,archive=True,repaircr=True,pelite=.05,cdamp=.1,sigma=.07) :159-164
     * SHADE(mfev,npinit,tol,archive=True,repaircr=True,h=100,npmin=4)            :166-171 */
    int archive;
    int repaircr;
    double pelite;
    double cdamp;
    double jade_sigma;
    int h;
    int npmin;
    /* APSO(mfev,tol,np,correct=True)                                     :265-269 */
    int correct;
    /* IPopCMAES(base,mfev,print=False,sigma0=2,nipop=True,ksigmadec=1.6,boundlambda=True) :137-142
     * BiPopCMAES(base,mfev,print=False,sigma0=2,maxlargeruns=9,nbipop=True,
     *            ksigmadec=1.6,kbudget=2)                                        :144-151
     * (the `base` optimizer is passed to bbo_create_restart) */
    int print;
    int nipop;             /* IPOP `nipop` / BIPOP `nbipop`                       */
    double ksigmadec;
    int boundlambda;
    int maxlargeruns;
    double kbudget;
    /* ---- extensions (no reference counterpart; defaults keep reference behaviour) */
    uint64_t seed;         /* Philox key.  The reference seeds from random_device +
                              clock and has no seed API (random.hpp:150-163)      */
    int device;            /* HIP device ordinal                                  */
    int populations;       /* independent populations advanced in lockstep by one
                              handle (>= 1); population p uses sub-stream p       */
    int poll_every;        /* generations between host polls of the device stop
                              flag inside bbo_optimize / bbo_run (default 8)      */
    /* ---- appended in round 1 (SepCMAES): keeps the layout of everything above */
    int adjustlr;          /* SepCmaes `adjustlr` (sep_cmaes.cpp:60-62)            */
    /* SANSDE(mfev,np,tol,repaircr=True,crref=5,pupdate=50,crupdate=25)  :174-177 */
    int crref;             /* generations between redraws of the per-individual CR */
    int pupdate;           /* generations between updates of the strategy probability */
    int crupdate;          /* generations between updates of the CR mean and of fp */
    /* CSO(mfev,stol,np,pcompete=3,ring=False,correct=True,vmax=0.2)     :272-275
     * (`stol` travels in `tol`, `correct` is shared with APSO) */
    int pcompete;          /* particles per competition                            */
    int ring;              /* ring neighbourhood instead of the swarm mean         */
    double vmax;           /* velocity clamp as a fraction of the box width        */
    /* CCPSO(mfev,sigmatol,np,pps,npps,correct=True,pcauchy=-1,local=None,localfreq=10) :291-295
     * (`sigmatol` travels in `tol`; the optional local optimizer: bbo_ccpso_set_local below) */
    int npps;              /* number of candidate swarm sizes                      */
    int pps[16];           /* the candidate swarm sizes (each must divide n)       */
    double pcauchy;        /* fixed Cauchy rate in (0,1), else adaptive            */
} bbo_params;

void bbo_params_default(bbo_params *p, int algo);

/* ---- life cycle ----------------------------------------------------------------
 * bbo_create            <- the Python constructors (py/multivariate_py.cpp:103-171,265-269)
 * bbo_create_restart    <- IPopCmaes / BiPopCmaes constructors taking `BaseCmaes *base`
 *                          (ipop_cmaes.cpp:56, bipop_cmaes.cpp:54).  `base` stays owned
 *                          by the caller and must outlive the driver, as in the reference.
 * bbo_destroy           <- destructor                                                  */
int bbo_create(const bbo_params *params, bbo_handle *out);
int bbo_create_restart(const bbo_params *params, bbo_handle base, bbo_handle *out);
int bbo_destroy(bbo_handle h);

/* ---- the four methods of MultivariateOptimizer (multivariate.h:132-146) ----------
 * lower/upper/guess are borrowed for the call and copied, like base_cmaes.cpp:63-64,124.
 * With populations > 1, guess holds populations*n doubles (one row per population).
 *
 * bbo_init      <- init(const multivariate_problem&, const double *guess)
 *                  Python: initialize(f, lower, upper, guess)  multivariate_py.cpp:397-416
 * bbo_iterate   <- iterate()                                   multivariate_py.cpp:418
 * bbo_solution  <- solution() -> {x, n_evals, converged}       multivariate_py.cpp:360-371,420
 *                  (population 0; bbo_solution_of for the others)
 * bbo_optimize  <- optimize(problem, guess) = init + loop      multivariate_py.cpp:376-395 */
int bbo_init(bbo_handle h, int n, const double *lower, const double *upper,
        const double *guess, const bbo_objective *objective);
int bbo_iterate(bbo_handle h);
int bbo_solution(bbo_handle h, double *x_out, int *n_evals, int *converged);
int bbo_solution_of(bbo_handle h, int population, double *x_out, int *n_evals,
        int *converged);
int bbo_optimize(bbo_handle h, int n, const double *lower, const double *upper,
        const double *guess, const bbo_objective *objective, double *x_out, int *n_evals,
        int *converged);

/* ---- throughput entry point (extension) -------------------------------------------
 * Runs up to max_generations generations without returning to the caller, stopping
 * early when every population's own stop rule has fired (the loop body of
 * base_cmaes.cpp:166-172 / shade.cpp:247-253 / apso.cpp:118-124 kept on the device).
 * generations_done counts launched generations; a population that stopped earlier is
 * frozen exactly at its stopping generation. */
int bbo_run(bbo_handle h, int max_generations, int *generations_done);

/* ---- state access (parity tests, checkpointing) ------------------------------------
 * Named read/write access to the optimizer state, the counterpart of the reference's
 * protected members (base_cmaes.h:52-63, cmaes.h:42-46, shade.h:44-51, apso.h:46-66).
 * bbo_get returns the element count (>= 0) or a negative bbo_status; call with cap 0
 * to size a buffer.  Keys are listed in HISTORY.md (last section) and DESIGN.md section 9.
 * One key changes semantics the reference's user can see: APSO "chunk" -- the number of particles
 * that move between two refreshes of the swarm's best inside a generation (the reference refreshes
 * after every particle, apso.cpp:194-197; the default here is np / 16 in whole workgroups, at least
 * 64; 0 = the whole swarm sees the best of the generation start). */
int bbo_get(bbo_handle h, const char *key, int population, double *out, int cap);
int bbo_set(bbo_handle h, const char *key, int population, const double *in, int count);

/* One phase of a CMA-ES generation, for step-level parity tests
 * (the template-method hooks of base_cmaes.h:80-92). */
typedef enum {
    BBO_PHASE_SAMPLE_EVALUATE = 0, /* samplePopulation + objective       cmaes.cpp:65-80       */
    BBO_PHASE_RANK = 1,            /* sort part of evaluateAndSortPopulation base_cmaes.cpp:221 */
    BBO_PHASE_UPDATE = 2,          /* updateDistribution without eigen   active_cmaes.cpp:71-164 */
    BBO_PHASE_EIGEN = 3,           /* eigenDecomposition                 cmaes.cpp:229-283     */
    BBO_PHASE_HISTORY_STOP = 4     /* updateHistory, it++, converged()   base_cmaes.cpp:191-209, cmaes.cpp:151-227 */
} bbo_cma_phase;
int bbo_cma_phase_run(bbo_handle h, int phase);

/* Injects the standard normals of the next sampling phase (populations*lambda*n doubles,
 * row-major) instead of drawing them on the device; used by the parity tests to feed the
 * reference's own draws.  Pass NULL to return to the device generator. */
int bbo_cma_inject_normals(bbo_handle h, const double *z, int count);

/* BaseCmaes::setParams(np, sigma, mfev) (base_cmaes.cpp:136-148): what the reference's restart
 * drivers call on their borrowed `base` before every inner run (bipop_cmaes.cpp:83,220,251;
 * ipop_cmaes.cpp:92,140).  Like the reference it switches `bound` off with a warning on stderr.
 * Takes effect at the next bbo_init / bbo_optimize; B and C keep their off-diagonals across
 * re-inits of one handle with the same n (cmaes.cpp:53-59). */
int bbo_cma_set_params(bbo_handle h, int np, double sigma0, int mfev);

/* (extension) a new Philox key for the next bbo_init / bbo_optimize of this handle; the
 * reference has no seed API (random.hpp:150-163). */
int bbo_cma_set_seed(bbo_handle h, uint64_t seed);

/* One evaluation of the handle's objective at x (n doubles), with the objective bound by the
 * last bbo_init / bbo_optimize: the restart drivers' re-evaluation of the point an inner run
 * returned (`_f._f(&x[0])`, bipop_cmaes.cpp:86,223,254; ipop_cmaes.cpp:95,143). */
int bbo_cma_evaluate(bbo_handle h, const double *x, double *f_out);

/* ---- CCPSO swarm groups sharded over GPUs (extension) --------------
 * The 2 (n/s) np context-vector evaluations of a CCPSO generation (CCPSOSearch::updateSwarm,
 * ccpso.cpp:241-260, each through evaluate :152-171) are independent while yhat is frozen.
 * With bbo_ccpso_set_shard(rank, world) a handle evaluates only its block of swarms,
 *   [nswarm * rank / world, nswarm * (rank + 1) / world),
 * and a generation becomes: phase 0 (regroup + evaluate this block) on every rank, ONE
 * all-gather of the fitness records, bbo_ccpso_merge_tables, phase 1 (the rest of updateSwarm,
 * updatePosition, the stop test: replicated, identical on every rank).  A record is
 * bbo_ccpso_table_record(h) doubles: only the rows of the swarms this rank evaluated, fX block
 * | fY block, each ceil(max swarms / world) * np doubles (max swarms = n / the smallest swarm
 * size); `device_memory` != 0 says the caller's pointer is device memory (e.g. an RCCL buffer),
 * else host memory.  One population.  While world > 1, bbo_iterate / bbo_run / bbo_optimize
 * return BBO_ERR_STATE: only the phase / merge protocol advances a sharded handle.
 * BUFFER LIFETIME with device_memory != 0: bbo_ccpso_merge_tables only ENQUEUES the merge kernel
 * on the handle's stream and returns; `gathered` must stay allocated and unmodified until the
 * next call on this handle that waits for the stream -- bbo_ccpso_export_tables (host or device),
 * bbo_get, bbo_solution, or bbo_ccpso_phase(h, 1) -- has returned (ShardedCCPSO reuses one
 * buffer per handle and overwrites it only in the next generation's all-gather, which follows the
 * next export).  A kernel error of the merge surfaces at that call.  With host memory the call
 * has copied `gathered` when it returns.  bbo_ccpso_export_tables with device_memory != 0 has
 * finished writing `dst` when it returns (the collective may start at once). */
int bbo_ccpso_set_shard(bbo_handle h, int rank, int world);
/* CCPSO's optional local optimizer (CCPSOSearch(..., local, localfreq), ccpso.cpp:51-70; used at
 * :116-118 and :371-435; py/multivariate_py.cpp:291-295).  `local` is BORROWED, like the
 * reference's pointer (and like `base` of bbo_create_restart): it must outlive h or be detached
 * with bbo_ccpso_set_local(h, NULL, 0).  After generation g with g % localfreq == 0, bbo_iterate /
 * bbo_run / bbo_optimize optimize one weight per swarm with it (objective evaluated on the host:
 * the callback of h's objective, or the built-in's formula), replace the context vector if the
 * result is better and charge the evaluations.  A CMA-ES `local` starts every search afresh
 * (B = C = I, seed = its own seed + search index).  One population; not with bbo_ccpso_set_shard. */
int bbo_ccpso_set_local(bbo_handle h, bbo_handle local, int localfreq);
int bbo_ccpso_phase(bbo_handle h, int phase);
int bbo_ccpso_table_record(bbo_handle h);
int bbo_ccpso_export_tables(bbo_handle h, double *dst, int device_memory);
int bbo_ccpso_merge_tables(bbo_handle h, const double *gathered, int world, int device_memory);

const char *bbo_last_error(bbo_handle h);   /* h may be NULL: last creation error */
const char *bbo_version(void);
int bbo_device_count(void);

#ifdef __cplusplus
}
#endif

#endif /* BBOPT_HIP_H_ */
