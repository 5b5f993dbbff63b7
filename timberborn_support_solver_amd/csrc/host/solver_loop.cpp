#include "solver_loop.hpp"

#include <chrono>
#include <stdexcept>

namespace tbs {

SolverResult run_solver(const Cnf& cnf, const mi355sat_opts* opts, std::vector<int8_t>& model,
                        mi355sat_stats_t& stats, const std::function<void(mi355sat*)>& on_interrupter) {
    mi355sat* s = mi355sat_new(opts);  // GlucoseSimp::default(), main.rs:295
    if (!s) throw std::runtime_error(std::string("Failed to create solver: ") + mi355sat_last_error(nullptr));
    auto fail = [&](const char* ctx) {
        std::string m = std::string(ctx) + ": " + mi355sat_last_error(s);
        mi355sat_free(s);
        throw std::runtime_error(m);
    };
    if (mi355sat_add_cnf(s, cnf.lits.data(), cnf.offsets.data(), cnf.n_clauses()) < 0) fail("Failed to add CNF");
    mi355sat_reserve(s, cnf.n_vars);
    if (on_interrupter) on_interrupter(s);   // solver.interrupter(), solver_runner.rs:13
    int rc = mi355sat_solve(s);              // solver_runner.rs:16
    if (rc < 0) fail("solve");
    model.assign(cnf.n_vars, 0);
    if (rc == MI355SAT_SAT && mi355sat_model(s, model.data(), cnf.n_vars) < 0) fail("full_solution");
    mi355sat_stats(s, &stats);
    if (on_interrupter) on_interrupter(nullptr);  // the handle is about to die
    mi355sat_free(s);
    return (SolverResult)rc;
}

std::vector<LoopIteration> solver_loop(const WorldGrid& world, const Encoding& encoding, PlatformLimits limits,
                                       const mi355sat_opts* opts,
                                       const std::function<void(const std::string&)>& out,
                                       const std::function<void(mi355sat*)>& on_interrupter) {
    std::vector<LoopIteration> hist;
    const Dims one{1, 1};
    for (;;) {
        Cnf cnf = encoding.with_limits(limits).into_cnf();   // main.rs:292-293
        LoopIteration it;
        auto lim = limits.card_limits.find(one);
        it.k = lim == limits.card_limits.end() ? (size_t)-1 : lim->second;
        std::vector<int8_t> model;
        auto t0 = std::chrono::steady_clock::now();
        it.result = run_solver(cnf, opts, model, it.stats, on_interrupter);
        it.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (it.result == SolverResult::Unsat) {
            out("No solution found for the current constraints");   // main.rs:332
            hist.push_back(std::move(it));
            return hist;
        }
        if (it.result == SolverResult::Interrupted) {
            out("Solver interrupted");                               // main.rs:336
            hist.push_back(std::move(it));
            return hist;
        }
        it.layout = PlatformLayout::from_assignment(model.data(), encoding.instance().n_vars, encoding);
        it.count = it.layout.platform_count();
        if (it.count == 0) {
            out("Found a solution with no platforms - aborting");   // main.rs:342
            hist.push_back(std::move(it));
            return hist;
        }
        limits.card_limits[one] = it.count - 1;                      // main.rs:346
        out("Solution found (" + std::to_string(it.count) + " platforms total)");
        for (auto& kv : it.layout.platform_stats())
            out(std::to_string(kv.first.w) + "x" + std::to_string(kv.first.h) + ": " + std::to_string(kv.second));
        it.valid = it.layout.validate(world).is_valid();
        out(it.valid ? "Solution validation OK" : "Solution validation FAILED");
        hist.push_back(std::move(it));
    }
}

}  // namespace tbs
