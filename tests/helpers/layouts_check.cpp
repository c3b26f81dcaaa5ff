// CPU check of the shipped decoder layouts (ria_amd/csrc/core_layouts.inc): each must unflatten and pass
// validate_core_tables() against the H this build generates; a corrupted copy must be rejected; a freshly annealed
// layout must validate too.  Prints one line per rate: "<rate> <ok> <cost_after> <floor> <rejects_corruption>".
#include <cstdio>
#include "../../ria_amd/csrc/host_tables.hpp"
int main() {
    int bad = 0;
    for (int rate = 0; rate < 6; ++rate) {
        const ria::LdpcCode c = ria::build_ldpc(rate);
        ria::CoreTables t;
        const bool ok = ria::load_saved_core_tables(c, t);
        int rejects = 0, tried = 0;
        if (ok) {
            // corrupt REAL edges (a padded slot may read any copy of its constant word, so a neighbouring copy is still valid);
            // and push one padded read out of its constant block
            auto real_row = [&](size_t pos) { while (t.row_addr[pos] / 4 >= t.big_word) pos = (pos + 1) % t.row_addr.size(); return pos; };
            auto real_col = [&](size_t pos) { while (t.col_addr[pos] / 4 >= t.tot_word) pos = (pos + 1) % t.col_addr.size(); return pos; };
            for (size_t pos : {size_t(3), t.row_addr.size() / 2, t.row_addr.size() - 1}) {
                ria::CoreTables u = t; u.row_addr[real_row(pos)] ^= 4; ++tried; rejects += !ria::validate_core_tables(c, u);
            }
            for (size_t pos : {size_t(1), t.col_addr.size() / 3}) {
                ria::CoreTables u = t; u.col_addr[real_col(pos)] += 4; ++tried; rejects += !ria::validate_core_tables(c, u);
            }
            for (size_t pos = 0; pos < t.col_addr.size(); ++pos)
                if (t.col_addr[pos] / 4 >= t.zero_word) { ria::CoreTables u = t; u.col_addr[pos] = static_cast<uint16_t>(4 * (t.zero_word + 64)); ++tried; rejects += !ria::validate_core_tables(c, u); break; }
            { ria::CoreTables u = t; std::swap(u.check_at[0], u.check_at[1]); ++tried; rejects += !ria::validate_core_tables(c, u); }
        }
        const ria::CoreTables fresh = ria::build_core_tables(c, 2000);
        const bool fresh_ok = ria::validate_core_tables(c, fresh);
        printf("%d %d %d %d %d/%d %d\n", rate, ok ? 1 : 0, t.conflict_cost_after, t.conflict_floor, rejects, tried, fresh_ok ? 1 : 0);
        bad += !ok || rejects != tried || !fresh_ok;
    }
    return bad ? 1 : 0;
}
