// options.h -- the library's test and tuning knobs as ONE explicit, thread-safe table (include/mcmcdate_mvn.h: mcd_set_option).
// Until round 3 these were environment variables read with getenv() on every call of the hot path: a stray variable silently changed
// which launch structure (and which rounding) a run took, and getenv() races with setenv() in a threaded host -- the reference runs
// `-threaded -N` (mcmc-date.cabal:42-43).  Now: an atomic int per knob, MCD_OPT_UNSET by default; the environment is consulted ONCE, when
// the library is loaded, to seed the table (so `MCD_SPLIT=0 python tools/...` still works for the measuring scripts), never afterwards.
#pragma once

namespace mcd {

enum Option {
    OPT_MH_PER_PHASE = 0,   // 1: two launches per step also where a whole-schedule kernel exists
    OPT_MH_SEGMENTS,        // 0: no segment kernels (dense or sparse likelihood)
    OPT_MH_INCREMENTAL,     // 0: every proposal by a full evaluation of the likelihood
    OPT_MH_PRIOR,           // 0: the ln prior inside the step kernel everywhere
    OPT_MH_PRIOR_CACHE,     // 0: every summand of the ln prior at every step (k_mh_step_wg)
    OPT_MH_STEP_WG,         // 1 / 0: force / forbid the workgroup-per-chain step kernel
    OPT_MH_CHAIN_LW,        // 0: the small-tree kernel with one wave per chain
    OPT_MH_INC_SLOTS,       // moved distances up to which a proposal counts as sparse (read by mcd_mh_create*)
    OPT_MH_SPARSE_SLOTS,    // the same for the streaming chain kernel (read by mcd_mh_create)
    OPT_SPLIT,              // 1: the row-split form wherever possible, 0: never
    OPT_SPLIT_G,            // force the number of row groups
    OPT_SPLIT_SCATTER,      // a tile's row groups on different XCDs (tests)
    OPT_SPLIT_NOROT,
    OPT_SPLIT_PROBE,        // timing probes (results are then garbage)
    OPT_GEOM,               // 21 | 41 | 42: compute waves, chains per wave of the column sweep
    OPT_WIDE_CT,            // 1 | 2 | 4: chains per workgroup / 16 of the multiply form
    OPT_SPARSE_QUAD,        // 1 / 0: force / forbid the one-launch form of the sparse log-density
    OPT_MH_PRIOR_WAVES,     // 0: the segment kernels without their prior waves (the chain wave evaluates the whole ln prior)
    OPT_LOADERS,            // 2: two loader waves in the column sweep's 512-chain geometry at 129 .. 256 dimensions instead of four (A/B)
    OPT_MH_SEG_TAIL,        // 0: a dense proposal that follows a segment is proposed by a launch of the step kernel, not by the segment's launch (A/B, tests)
    OPT_MH_AHEAD_FROM,      // nodes from which a segment kernel's chain wave draws the next proposal ahead of the decision (default: kSegAheadFrom)
    OPT_MH_PRIOR_DRAWS,     // 0: the next step's proposal is not drawn by a prior wave (the chain wave draws ahead itself from MCD_MH_AHEAD_FROM nodes; A/B, tests)
    OPT_COUNT
};
constexpr int MCD_OPT_UNSET = -2147483647 - 1;

int opt_get(Option o);                                   // MCD_OPT_UNSET or the value
// reporting: the dynamic LDS (bytes per workgroup) of the persistent Metropolis-Hastings kernel launched last by this thread -- the
// profiler's kernel trace shows the static group segment only (0 for these kernels); mcd_mh_last_dynamic_lds reads it after a run
void note_dynamic_lds(unsigned long long bytes);
unsigned long long last_dynamic_lds();
inline bool opt_is(Option o, int v) { return opt_get(o) == v; }
inline int opt_or(Option o, int dflt) { const int v = opt_get(o); return v == MCD_OPT_UNSET ? dflt : v; }

}  // namespace mcd
