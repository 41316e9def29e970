// Instantiations of the block-split median kernel (windows up to 21): smh_median_split.h.
#include "smh_median_split.h"

namespace smh_median {

#define SMH_SPLIT_E2(a, b) {a, b, SplitCfg<a, b>::kThreads, hpss_median_split_kernel<a, b>},
#define SMH_SPLIT_SINGLE(w) SMH_SPLIT_E2(w, 0) SMH_SPLIT_E2(0, w)

const SplitEntry kSplit[] = {
    // both filters in one launch: the reference's configuration, BASELINE config 2, the small end of the sweep
    SMH_SPLIT_E2(21, 11) SMH_SPLIT_E2(17, 17) SMH_SPLIT_E2(11, 11) SMH_SPLIT_E2(11, 21) SMH_SPLIT_E2(21, 21)
    // single filters
    SMH_SPLIT_SINGLE(3) SMH_SPLIT_SINGLE(5) SMH_SPLIT_SINGLE(7) SMH_SPLIT_SINGLE(9) SMH_SPLIT_SINGLE(11)
    SMH_SPLIT_SINGLE(13) SMH_SPLIT_SINGLE(15) SMH_SPLIT_SINGLE(17) SMH_SPLIT_SINGLE(19) SMH_SPLIT_SINGLE(21)};

const SplitEntry *find_split_kernel(int lh, int lp) {
    for (const SplitEntry &e : kSplit)
        if (e.lh == lh && e.lp == lp) return &e;
    return nullptr;
}

#define SMH_PERSIST_E3(a, b, t) {a, b, t, hpss_median_persist_kernel<a, b, t>},
#define SMH_PERSIST_E2(a, b) SMH_PERSIST_E3(a, b, 512) SMH_PERSIST_E3(a, b, 768)
const PersistEntry kPersist[] = {SMH_PERSIST_E2(21, 11) SMH_PERSIST_E2(17, 17) SMH_PERSIST_E2(11, 11)
                                 SMH_PERSIST_E2(11, 21) SMH_PERSIST_E2(21, 21) SMH_PERSIST_E3(17, 17, 1024)};

const PersistEntry *find_persist_kernel(int lh, int lp, int threads) {
    for (const PersistEntry &e : kPersist)
        if (e.lh == lh && e.lp == lp && e.threads == threads) return &e;
    return nullptr;
}

}  // namespace smh_median
