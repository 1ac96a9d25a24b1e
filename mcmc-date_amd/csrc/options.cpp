// options.cpp -- see options.h.  mcd_set_option / mcd_get_option of include/mcmcdate_mvn.h.
#include "options.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../include/mcmcdate_mvn.h"

extern "C" int mcd_set_last_error_(int code, const char* msg);

namespace mcd {
namespace {

const char* const kNames[OPT_COUNT] = {
    "MCD_MH_PER_PHASE", "MCD_MH_SEGMENTS", "MCD_MH_INCREMENTAL", "MCD_MH_PRIOR", "MCD_MH_PRIOR_CACHE", "MCD_MH_STEP_WG", "MCD_MH_CHAIN_LW",
    "MCD_MH_INC_SLOTS", "MCD_MH_SPARSE_SLOTS", "MCD_SPLIT", "MCD_SPLIT_G", "MCD_SPLIT_SCATTER", "MCD_SPLIT_NOROT", "MCD_SPLIT_PROBE", "MCD_GEOM",
    "MCD_WIDE_CT", "MCD_SPARSE_QUAD", "MCD_MH_PRIOR_WAVES", "MCD_LOADERS", "MCD_MH_SEG_TAIL", "MCD_MH_AHEAD_FROM", "MCD_MH_PRIOR_DRAWS"};

struct Table {
    std::atomic<int> v[OPT_COUNT];
    Table()
    {
        for (int i = 0; i < OPT_COUNT; ++i) {
            const char* e = std::getenv(kNames[i]);        // once, at load time: the seed of the table
            v[i].store((e && *e) ? std::atoi(e) : MCD_OPT_UNSET, std::memory_order_relaxed);
        }
    }
};
Table& table()
{
    static Table t;
    return t;
}
const Table& g_seed_at_load = table();                     // (constructed when the library is loaded, not at first use)

int find(const char* name)
{
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kNames[i]) == 0) return i;
    return -1;
}

}  // namespace

int opt_get(Option o) { return table().v[o].load(std::memory_order_relaxed); }

namespace {
thread_local unsigned long long t_last_lds = 0;
}
void note_dynamic_lds(unsigned long long bytes) { t_last_lds = bytes; }
unsigned long long last_dynamic_lds() { return t_last_lds; }

}  // namespace mcd

extern "C" {

int mcd_set_option(const char* name, const char* value)
{
    const int i = mcd::find(name);
    if (i < 0) return mcd_set_last_error_(MCD_ERR_INVALID_ARG, "mcd_set_option: unknown option");
    mcd::table().v[i].store((value && *value) ? std::atoi(value) : mcd::MCD_OPT_UNSET, std::memory_order_relaxed);
    return MCD_OK;
}

int mcd_get_option(const char* name, int* is_set, int* value)
{
    const int i = mcd::find(name);
    if (i < 0) return mcd_set_last_error_(MCD_ERR_INVALID_ARG, "mcd_get_option: unknown option");
    const int v = mcd::table().v[i].load(std::memory_order_relaxed);
    if (is_set) *is_set = (v != mcd::MCD_OPT_UNSET) ? 1 : 0;
    if (value) *value = (v != mcd::MCD_OPT_UNSET) ? v : 0;
    return MCD_OK;
}

}  // extern "C"
