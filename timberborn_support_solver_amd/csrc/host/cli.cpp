// tbs_cli — minimal command-line driver of the solve path (NOT the reference's REPL/UI):
//   tbs_cli rect W H [-l1:K] [--platforms default|1x1] [--workers N] [--sweep] [--gpu N] [--seed N] [--verbose] [--no-simp] [--eliminate]
//   tbs_cli file PATH.toml [-l1:K] ...
// Mirrors `solve -l<dims>:<n>` of crates/repl/src/main.rs:44-75,248-261: encode once, then solver_loop.
// Ctrl-C calls mi355sat_interrupt (main.rs:297-324).
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "solver_loop.hpp"

static std::atomic<mi355sat*> g_current{nullptr};
static std::atomic<int> g_interrupted{0};
static void on_sigint(int) {
    g_interrupted.store(1);
    mi355sat* s = g_current.load();
    if (s) mi355sat_interrupt(s);
}

int main(int argc, char** argv) {
    using namespace tbs;
    try {
        if (argc < 3) {
            fprintf(stderr, "usage: %s rect W H | file PATH [-l<dims>:<n>]... [--platforms default|1x1] [--workers N] [--sweep] [--gpu N] [--seed N] "
                            "[--verbose] [--no-simp] [--eliminate]\n", argv[0]);
            return 2;
        }
        WorldGrid grid;
        int a = 1;
        if (!strcmp(argv[a], "rect") && argc >= 4) { grid = WorldGrid::rect(atoi(argv[a + 1]), atoi(argv[a + 2])); a += 3; }
        else if (!strcmp(argv[a], "file")) { grid = WorldGrid::from_toml_file(argv[a + 1]); a += 2; }
        else throw std::runtime_error("expected `rect W H` or `file PATH`");
        std::vector<Dims> defs = platforms_default();
        PlatformLimits limits;
        mi355sat_opts opts{};
        opts.device = -1;
        bool sweep = false;
        for (; a < argc; a++) {
            std::string arg = argv[a];
            if (arg.rfind("-l", 0) == 0) {             // -l<dims>:<n>, dims = AxB or A (=AxA), main.rs:120-142
                std::string kv = arg.substr(2);
                size_t c = kv.find(':');
                if (c == std::string::npos) throw std::runtime_error("missing/invalid delimiter");
                std::string d = kv.substr(0, c);
                size_t x = d.find('x');
                Dims dims = x == std::string::npos ? Dims{atoi(d.c_str()), atoi(d.c_str())}
                                                   : Dims{atoi(d.substr(0, x).c_str()), atoi(d.substr(x + 1).c_str())};
                limits.card_limits[dims] = (size_t)atol(kv.substr(c + 1).c_str());
            } else if (arg == "--platforms" && a + 1 < argc) {
                if (!strcmp(argv[++a], "1x1")) defs = {Dims{1, 1}};
            } else if (arg == "--workers" && a + 1 < argc) opts.workers = atoi(argv[++a]);
            else if (arg == "--gpu" && a + 1 < argc) opts.device = atoi(argv[++a]);          // HIP device ordinal (one process per GPU)
            else if (arg == "--seed" && a + 1 < argc) opts.seed = strtoull(argv[++a], nullptr, 10);   // diversification seed
            else if (arg == "--verbose") opts.verbose = 1;
            else if (arg == "--no-simp") opts.simp = -1;
            else if (arg == "--eliminate") opts.simp = 2;     // + bounded variable elimination before every search
            else if (arg == "--sweep") sweep = true;   // the bounds below the first one as one batch on the device
            else throw std::runtime_error("unknown argument " + arg);
        }
        Encoding enc = Encoding::encode(defs, grid);
        signal(SIGINT, on_sigint);
        auto print = [](const std::string& l) { std::cout << l << std::endl; };
        auto hist = sweep ? solver_loop_sweep(grid, enc, limits, &opts, print, [](mi355sat* s) { g_current.store(s); }, &g_interrupted)
                          : solver_loop(grid, enc, limits, &opts, print, [](mi355sat* s) { g_current.store(s); });
        return hist.empty() ? 1 : 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
