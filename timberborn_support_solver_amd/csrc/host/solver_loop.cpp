#include "solver_loop.hpp"

#include <algorithm>
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
                                       const std::function<void(mi355sat*)>& on_interrupter, size_t max_iterations) {
    std::vector<LoopIteration> hist;
    const Dims one{1, 1};
    while (hist.size() < max_iterations) {
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
    return hist;
}

std::vector<LoopIteration> solver_loop_sweep(const WorldGrid& world, const Encoding& encoding, const PlatformLimits& limits,
                                             const mi355sat_opts* opts,
                                             const std::function<void(const std::string&)>& out,
                                             const std::function<void(mi355sat*)>& on_interrupter,
                                             const std::atomic<int>* interrupted) {
    const Dims one{1, 1};
    if (limits.card_limits.size() != 1 || !limits.card_limits.count(one) || !limits.weights.empty() || limits.has_weight_limit)
        throw std::runtime_error("solver_loop_sweep handles a single 1x1 cardinality limit; use solver_loop");
    std::vector<LoopIteration> hist = solver_loop(world, encoding, limits, opts, out, on_interrupter, 1);
    if (hist.empty() || hist.back().result != SolverResult::Sat || hist.back().count == 0) return hist;
    const size_t k0 = hist.back().count - 1;
    PlatformLimits below;
    below.card_limits[one] = k0;
    std::vector<std::vector<int32_t>> outs;
    Cnf cnf = encoding.with_limits(below).into_cnf(&outs);
    const std::vector<int32_t> card = outs.empty() ? std::vector<int32_t>{} : outs[0];
    std::vector<size_t> ks;
    std::vector<int32_t> assumps;
    std::vector<uint64_t> offs{0};
    for (size_t k = k0 + 1; k-- > 0;) {
        ks.push_back(k);
        if (k < card.size()) assumps.push_back(-card[k]);     // at most k  <=>  not (at least k+1)
        offs.push_back(assumps.size());
    }
    mi355sat* s = mi355sat_new(opts);
    if (!s) throw std::runtime_error(std::string("Failed to create solver: ") + mi355sat_last_error(nullptr));
    auto fail = [&](const char* ctx) {
        std::string m = std::string(ctx) + ": " + mi355sat_last_error(s);
        if (on_interrupter) on_interrupter(nullptr);
        mi355sat_free(s);
        throw std::runtime_error(m);
    };
    if (mi355sat_add_cnf(s, cnf.lits.data(), cnf.offsets.data(), cnf.n_clauses()) < 0) fail("Failed to add CNF");
    mi355sat_reserve(s, cnf.n_vars);
    if (on_interrupter) on_interrupter(s);
    auto t0 = std::chrono::steady_clock::now();
    if (mi355sat_sweep_begin(s, assumps.empty() ? nullptr : assumps.data(), offs.data(), ks.size()) < 0) fail("sweep_begin");
    std::vector<int32_t> res(ks.size(), 0);
    std::vector<char> looked(ks.size(), 0);
    std::vector<int8_t> model(cnf.n_vars), best_model;
    long best_c = -1, unsat_k = -1;
    bool stopped = false, specialize = false;
    for (;;) {
        uint64_t nd = 0;
        if (mi355sat_sweep_step(s, res.data(), &nd) < 0) fail("sweep_step");
        for (size_t i = 0; i < ks.size(); i++) {
            if (res[i] == MI355SAT_UNSAT) unsat_k = std::max(unsat_k, (long)ks[i]);
            else if (res[i] == MI355SAT_SAT && !looked[i]) {
                looked[i] = 1;
                if (mi355sat_sweep_model_of(s, i, model.data(), cnf.n_vars) < 0) fail("full_solution");
                long c = (long)PlatformLayout::from_assignment(model.data(), encoding.instance().n_vars, encoding).platform_count();
                if (best_c < 0 || c < best_c) { best_c = c; best_model = model; }
            }
        }
        if (best_c >= 0 && (unsat_k + 1 >= best_c || best_c == 0)) break;
        if (nd == ks.size()) break;
        if (interrupted && interrupted->load()) { stopped = true; break; }
        std::vector<uint64_t> drop;
        for (size_t i = 0; i < ks.size(); i++)
            if (res[i] == 0 && ((best_c >= 0 && (long)ks[i] >= best_c) || (long)ks[i] < unsat_k)) drop.push_back(i);
        if (!drop.empty() && mi355sat_sweep_drop(s, drop.data(), drop.size()) < 0) fail("sweep_drop");
        // the two open bounds that decide the loop - the highest (a model there lowers the ceiling) and the lowest (a
        // refutation there raises the floor) - share the fleet; the ones in between keep a few workers each
        long hi = -1, lo = -1;
        for (size_t i = 0; i < ks.size(); i++)
            if (res[i] == 0 && (best_c < 0 || (long)ks[i] < best_c) && (long)ks[i] > unsat_k) {
                if (hi < 0 || (long)ks[i] > hi) hi = (long)ks[i];
                if (lo < 0 || (long)ks[i] < lo) lo = (long)ks[i];
            }
        // The batch is the fast way DOWN, not the fast way to the last refutation (a bound posed as an assumption
        // over a looser bound's totalizer is refuted much more slowly than with its own CNF: loop.py): after two
        // seconds the sequential loop finishes from the best count.
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { specialize = true; break; }
        std::vector<double> weights(ks.size(), 0.02);
        for (size_t i = 0; i < ks.size(); i++) if ((long)ks[i] == hi || (long)ks[i] == lo) weights[i] = 1.0;
        if (mi355sat_sweep_set_weights(s, weights.data(), weights.size()) < 0) fail("sweep_set_weights");
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    mi355sat_sweep_end(s);
    mi355sat_stats_t stats{};
    mi355sat_stats(s, &stats);
    if (on_interrupter) on_interrupter(nullptr);
    mi355sat_free(s);
    if (best_c >= 0) {
        LoopIteration it;
        it.k = k0; it.result = SolverResult::Sat; it.count = (size_t)best_c; it.seconds = dt; it.stats = stats;
        it.layout = PlatformLayout::from_assignment(best_model.data(), encoding.instance().n_vars, encoding);
        if (best_c == 0) {
            out("Found a solution with no platforms - aborting");
            hist.push_back(std::move(it));
            return hist;
        }
        out("Solution found (" + std::to_string(it.count) + " platforms total)");
        for (auto& kv : it.layout.platform_stats())
            out(std::to_string(kv.first.w) + "x" + std::to_string(kv.first.h) + ": " + std::to_string(kv.second));
        it.valid = it.layout.validate(world).is_valid();
        out(it.valid ? "Solution validation OK" : "Solution validation FAILED");
        hist.push_back(std::move(it));
    }
    if (specialize) {
        PlatformLimits rest;
        rest.card_limits[one] = (size_t)((best_c >= 0 ? best_c : (long)k0 + 1) - 1);
        std::vector<LoopIteration> tail = solver_loop(world, encoding, rest, opts, out, on_interrupter);
        for (auto& it : tail) hist.push_back(std::move(it));
        return hist;
    }
    LoopIteration last;
    last.seconds = dt; last.stats = stats;
    if (stopped) {
        last.k = best_c > 0 ? (size_t)(best_c - 1) : k0; last.result = SolverResult::Interrupted;
        out("Solver interrupted");
        hist.push_back(std::move(last));
    } else if (best_c < 0 || unsat_k + 1 >= best_c) {
        last.k = best_c < 0 ? (size_t)std::max(unsat_k, 0l) : (size_t)(best_c - 1); last.result = SolverResult::Unsat;
        out("No solution found for the current constraints");
        hist.push_back(std::move(last));
    }
    return hist;
}

}  // namespace tbs
