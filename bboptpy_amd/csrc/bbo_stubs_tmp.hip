#include "bbo_common.hpp"
namespace bbo {
Optimizer* make_restart_driver(const bbo_params &, Optimizer *) { throw Error(BBO_ERR_ARG, "restart driver not built yet"); }
}
