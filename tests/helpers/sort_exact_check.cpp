// Host check of ria_amd/csrc/sort_exact.hpp against the C++ library the reference is built with:
// the first `want` positions must equal std::sort's, ties included.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../ria_amd/csrc/sort_exact.hpp"

int main() {
    std::mt19937 rng(12345);
    int stack[192];
    long bad = 0, cases = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        int n = (trial < 64) ? trial : static_cast<int>(rng() % 2200);
        int levels = 1 + static_cast<int>(rng() % 40);   // few distinct keys -> many ties
        std::vector<ria::Suspect> a(n);
        for (int i = 0; i < n; ++i) { a[i].frame_bit = i; a[i].abs_llr = static_cast<float>(rng() % levels) * 0.25f; }
        if (trial % 7 == 0) std::sort(a.begin(), a.end(), [](auto& x, auto& y) { return x.abs_llr > y.abs_llr; });  // adversarial-ish
        std::vector<ria::Suspect> ref = a, full = a, part = a, heap = a, hp = a;
        std::sort(ref.begin(), ref.end(), ria::suspect_lt);
        ria::sort_exact_prefix(full.data(), n, n, stack, ria::suspect_lt);
        ria::sort_exact_prefix(part.data(), n, 30, stack, ria::suspect_lt);
        for (int i = 0; i < n; ++i) if (full[i].frame_bit != ref[i].frame_bit) { ++bad; break; }
        for (int i = 0; i < std::min(n, 30); ++i) if (part[i].frame_bit != ref[i].frame_bit) { ++bad; break; }
        {   // the data-parallel formulation (what the GPU wave runs): same prefix
            std::vector<ria::Suspect> lp = a, tmp(n + 64);
            std::vector<int> al(n + 1), bl(n + 1);
            ria::sort_exact_prefix_lists(lp.data(), n, 30, stack, ria::suspect_lt, al.data(), bl.data(), tmp.data());
            for (int i = 0; i < std::min(n, 30); ++i) if (lp[i].frame_bit != ref[i].frame_bit) { ++bad; break; }
            std::vector<ria::Suspect> lf = a;
            ria::sort_exact_prefix_lists(lf.data(), n, n, stack, ria::suspect_lt, al.data(), bl.data(), tmp.data());
            for (int i = 0; i < n; ++i) if (lf[i].frame_bit != ref[i].frame_bit) { ++bad; break; }
        }
        // forced depth budget 0: the library's fallback is __partial_sort(first, last, last)
        if (n > 16) {
            std::partial_sort(heap.begin(), heap.end(), heap.end(), ria::suspect_lt);
            ria::sort_exact_prefix(hp.data(), n, n, stack, ria::suspect_lt, 0);
            for (int i = 0; i < n; ++i) if (hp[i].frame_bit != heap[i].frame_bit) { ++bad; break; }
        }
        ++cases;
    }
    std::printf("cases %ld mismatches %ld\n", cases, bad);
    return bad ? 1 : 0;
}
