// solver_loop.hpp — C++ mirror of the reference's solver launch and refinement loop over the C ABI:
//   run_solver<S>   crates/repl/src/solver_runner.rs:8-20
//   solver_loop     crates/repl/src/main.rs:280-366
#pragma once
#include <atomic>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/mi355sat.h"
#include "tbs_host.hpp"

namespace tbs {

enum class SolverResult { Sat = 10, Unsat = 20, Interrupted = 0 };

struct LoopIteration {
    size_t k = 0;                 // bound on 1x1 (= total platforms) of this iteration
    SolverResult result = SolverResult::Interrupted;
    size_t count = 0;             // platform_count() of the layout (Sat only)
    bool valid = false;           // validate().is_valid()
    double seconds = 0;
    mi355sat_stats_t stats{};
    PlatformLayout layout;
};

// run_solver: fresh solver, add_cnf, hand the interrupter out, solve.  `on_interrupter` receives
// the handle whose mi355sat_interrupt() may be called from any thread while solve() runs.
SolverResult run_solver(const Cnf& cnf, const mi355sat_opts* opts, std::vector<int8_t>& model,
                        mi355sat_stats_t& stats, const std::function<void(mi355sat*)>& on_interrupter = {});

// solver_loop: repeat { with_limits -> into_cnf -> fresh solver -> solve -> layout -> k := count-1 }
// until Unsat / Interrupted / a layout without platforms; prints the reference's messages via `out`.
std::vector<LoopIteration> solver_loop(const WorldGrid& world, const Encoding& encoding, PlatformLimits limits,
                                       const mi355sat_opts* opts,
                                       const std::function<void(const std::string&)>& out,
                                       const std::function<void(mi355sat*)>& on_interrupter = {},
                                       size_t max_iterations = (size_t)-1);

// The same refinement as ONE batch on the device: every bound k0, k0-1, ..., 0 is an assumption
// set over one CNF built for k0; a SAT model with c platforms answers every bound >= c, an UNSAT bound every
// bound below it (those instances are withdrawn, mi355sat_sweep_drop).  Done when max UNSAT k + 1 == min
// count.  Prints the reference's messages for the iterations the sequential loop would still have to make
// (the best layout, then the refuting bound).  The first iteration (the loose start bound) is made exactly as
// the reference makes it; the batch covers count-1 .. 0.  Only the `-l1:K` form.
std::vector<LoopIteration> solver_loop_sweep(const WorldGrid& world, const Encoding& encoding, const PlatformLimits& limits,
                                             const mi355sat_opts* opts,
                                             const std::function<void(const std::string&)>& out,
                                             const std::function<void(mi355sat*)>& on_interrupter = {},
                                             const std::atomic<int>* interrupted = nullptr);

}  // namespace tbs
