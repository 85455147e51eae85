#include "bbo_common.hpp"
namespace bbo {
Optimizer* make_pso_engine(const bbo_params &) { throw Error(BBO_ERR_ARG, "PSO engine not built yet"); }
Optimizer* make_restart_driver(const bbo_params &, Optimizer *) { throw Error(BBO_ERR_ARG, "restart driver not built yet"); }
}
