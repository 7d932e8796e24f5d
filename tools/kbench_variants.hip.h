// kbench_variants.hip.h -- experimental variants of the hot kernels, timed side by side by tools/kbench.hip on one
// MI355X (methodology: one process, interleaved rounds).  Nothing here ships: a variant that wins is moved into
// mc_water_ls_mw_amd/csrc/ and the loser is deleted from this file.
#pragma once

namespace kb {
using namespace mw;
}  // namespace kb
