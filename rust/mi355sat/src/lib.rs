//! `rustsat::solvers::Solve` over libmi355sat.so - the same kind of wrapper rustsat-glucose is over its
//! vendored C++ solver (an IPASIR-shaped C API), so the reference swaps backends in two lines:
//!
//! ```text
//! - use rustsat_glucose::simp::Glucose as GlucoseSimp;     // crates/repl/src/main.rs:17, crates/gui/src/main.rs:2
//! + use mi355sat::Mi355Sat as GlucoseSimp;
//! ```
//!
//! Trait surface = what the reference exercises: `Default`, `add_cnf` / `add_clause_ref`,
//! `interrupter`, `solve`, `full_solution` (through `lit_val` + `max_var`), `stats`.
use std::os::raw::{c_char, c_int, c_void};

use rustsat::instances::Cnf;
use rustsat::solvers::{Interrupt, InterruptSolver, Solve, SolveStats, SolverResult, SolverStats};
use rustsat::types::{Cl, Clause, Lit, TernaryVal, Var};

/// Mirror of `mi355sat_opts` (include/mi355sat.h); zero = defaults.
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct Opts {
    pub device: i32, pub workers: i32, pub conflict_budget: i64, pub slice_conflicts: i32, pub seed: u64,
    pub verbose: i32, pub reduce_first: i32, pub reduce_inc: i32, pub lds_val: i32, pub max_groups: i32,
    pub slice_ms: i32, pub cube_split: i32, pub share: i32, pub share_lbd: i32, pub share_len: i32,
    pub share_interval: i32, pub var_order: i32, pub ramp: i32, pub one_per_simd: i32, pub simp: i32, pub phase_mix: i32, pub rephase: i32, pub restart_k_pct: i32, pub restart_k2_pct: i32, pub import_pct: i32, pub vivify: i32, pub rebalance: i32, pub deterministic: i32,
}

/// Mirror of `mi355sat_stats_t`.
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct Stats {
    pub propagations: u64, pub decisions: u64, pub conflicts: u64, pub restarts: u64, pub learnts: u64,
    pub learnt_literals: u64, pub reduce_dbs: u64, pub n_clauses: u64, pub max_var: u64, pub avg_clause_len: f64,
    pub solve_seconds: f64, pub kernel_seconds: f64, pub kernel_launches: u64, pub n_deq: u64, pub n_watch: u64,
    pub n_cl_lit: u64, pub n_move: u64, pub n_enq: u64, pub n_sat: u64, pub n_unsat: u64, pub n_terminated: u64,
    pub bcp_steps: u64, pub bcp_requeued: u64, pub shared_exported: u64, pub shared_imported: u64,
    pub shared_imported_units: u64, pub simp_units: u64, pub simp_equivalences: u64, pub simp_clauses_removed: u64, pub workers: u64, pub simp_eliminated: u64,
}

extern "C" {
    fn mi355sat_new(opts: *const Opts) -> *mut c_void;
    fn mi355sat_free(s: *mut c_void);
    fn mi355sat_abi_sizes(stats_size: *mut u64) -> u64;
    fn mi355sat_last_error(s: *const c_void) -> *const c_char;
    fn mi355sat_add_cnf(s: *mut c_void, lits: *const i32, offsets: *const u64, n: u64) -> c_int;
    fn mi355sat_add(s: *mut c_void, lit_or_0: i32) -> c_int;
    fn mi355sat_reserve(s: *mut c_void, n_vars: u64) -> c_int;
    fn mi355sat_solve(s: *mut c_void) -> c_int;
    fn mi355sat_val(s: *mut c_void, lit: i32) -> i32;
    fn mi355sat_interrupt(s: *mut c_void);
    fn mi355sat_stats(s: *const c_void, out: *mut Stats) -> c_int;
}

pub struct Mi355Sat { h: *mut c_void }
// The handle may move between OS threads between calls (solver_runner.rs:15 moves the solver into tokio's
// blocking pool); every entry point of the library binds its device itself.
unsafe impl Send for Mi355Sat {}

impl Mi355Sat {
    pub fn with_opts(opts: &Opts) -> anyhow::Result<Self> {
        let mut st_size = 0u64;
        let opt_size = unsafe { mi355sat_abi_sizes(&mut st_size) };
        anyhow::ensure!(opt_size as usize == std::mem::size_of::<Opts>() && st_size as usize == std::mem::size_of::<Stats>(),
                        "libmi355sat was built from another include/mi355sat.h than this crate mirrors");
        let h = unsafe { mi355sat_new(opts) };
        if h.is_null() {
            let m = unsafe { std::ffi::CStr::from_ptr(mi355sat_last_error(std::ptr::null())) };
            anyhow::bail!("mi355sat_new failed: {}", m.to_string_lossy());
        }
        Ok(Self { h })
    }
    fn err(&self) -> anyhow::Error {
        let m = unsafe { std::ffi::CStr::from_ptr(mi355sat_last_error(self.h)) };
        anyhow::anyhow!(m.to_string_lossy().into_owned())
    }
    fn raw_stats(&self) -> Stats {
        let mut st = Stats::default();
        unsafe { mi355sat_stats(self.h, &mut st) };
        st
    }
}

impl Default for Mi355Sat {                    // S::default(): main.rs:295, solver_backend.rs:79
    fn default() -> Self { Self::with_opts(&Opts { device: -1, ..Opts::default() }).expect("no usable HIP device") }
}
impl Drop for Mi355Sat { fn drop(&mut self) { unsafe { mi355sat_free(self.h) } } }

fn ipasir(l: Lit) -> i32 { let v = l.vidx32() as i32 + 1; if l.is_neg() { -v } else { v } }

impl Solve for Mi355Sat {
    fn signature(&self) -> &'static str { "mi355sat (HIP/gfx950 wave-parallel CDCL)" }
    fn add_clause_ref<C: AsRef<Cl> + ?Sized>(&mut self, c: &C) -> anyhow::Result<()> {
        for l in c.as_ref().iter() {
            if unsafe { mi355sat_add(self.h, ipasir(*l)) } < 0 { return Err(self.err()); }
        }
        if unsafe { mi355sat_add(self.h, 0) } < 0 { return Err(self.err()); }
        Ok(())
    }
    // Bulk override of the per-literal default (solver_runner.rs:12): one CSR hand-over, one FFI call.
    fn add_cnf(&mut self, cnf: Cnf) -> anyhow::Result<()> {
        let mut lits: Vec<i32> = Vec::new();
        let mut offsets: Vec<u64> = vec![0];
        for cl in cnf.iter() {
            lits.extend(cl.iter().map(|l| ipasir(*l)));
            offsets.push(lits.len() as u64);
        }
        if unsafe { mi355sat_add_cnf(self.h, lits.as_ptr(), offsets.as_ptr(), (offsets.len() - 1) as u64) } < 0 {
            return Err(self.err());
        }
        Ok(())
    }
    fn reserve(&mut self, max_var: Var) -> anyhow::Result<()> {
        if unsafe { mi355sat_reserve(self.h, max_var.idx() as u64 + 1) } < 0 { return Err(self.err()); }
        Ok(())
    }
    fn solve(&mut self) -> anyhow::Result<SolverResult> {            // solver_runner.rs:16
        match unsafe { mi355sat_solve(self.h) } {
            10 => Ok(SolverResult::Sat),
            20 => Ok(SolverResult::Unsat),
            0 => Ok(SolverResult::Interrupted),
            _ => Err(self.err()),                                     // negative codes -> anyhow error
        }
    }
    fn lit_val(&self, lit: Lit) -> anyhow::Result<TernaryVal> {      // full_solution(): main.rs:329, app.rs:154
        let l = ipasir(lit);
        Ok(match unsafe { mi355sat_val(self.h, l) } { v if v == l => TernaryVal::True, 0 => TernaryVal::DontCare, _ => TernaryVal::False })
    }
}

impl Extend<Clause> for Mi355Sat {
    fn extend<T: IntoIterator<Item = Clause>>(&mut self, it: T) { for c in it { self.add_clause_ref(&c).expect("add_clause") } }
}
impl<'a> Extend<&'a Clause> for Mi355Sat {
    fn extend<T: IntoIterator<Item = &'a Clause>>(&mut self, it: T) { for c in it { self.add_clause_ref(c).expect("add_clause") } }
}

/// `S::Interrupter`: `Send + 'static`, called through `&self` from another task while `solve()` runs
/// (main.rs:298-323).  It only sets a flag; the solver must outlive it, as with rustsat-glucose - true at both
/// call sites (the solver comes back from the blocking task, solver_runner.rs:15-17).
pub struct Interrupter(*mut c_void);
unsafe impl Send for Interrupter {}
unsafe impl Sync for Interrupter {}
impl InterruptSolver for Interrupter { fn interrupt(&self) { unsafe { mi355sat_interrupt(self.0) } } }     // main.rs:316
impl Interrupt for Mi355Sat {
    type Interrupter = Interrupter;
    fn interrupter(&mut self) -> Interrupter { Interrupter(self.h) }                                       // solver_runner.rs:13
}

impl SolveStats for Mi355Sat {                                                                            // main.rs:363, app.rs:148-151
    fn stats(&self) -> SolverStats {
        let st = self.raw_stats();
        SolverStats {
            n_sat: st.n_sat as usize, n_unsat: st.n_unsat as usize, n_terminated: st.n_terminated as usize,
            n_clauses: st.n_clauses as usize, max_var: (st.max_var > 0).then(|| Var::new(st.max_var as u32 - 1)),
            avg_clause_len: st.avg_clause_len as f32, cpu_solve_time: std::time::Duration::from_secs_f64(st.solve_seconds),
        }
    }
    fn max_var(&self) -> Option<Var> { let m = self.raw_stats().max_var; (m > 0).then(|| Var::new(m as u32 - 1)) }
}
