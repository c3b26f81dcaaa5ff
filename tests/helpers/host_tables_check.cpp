// Dumps the host-built Schmidl-Cox tables (ria_amd/csrc/host_tables.hpp) as raw float32 so that the CPU test can
// compare them with the reference's (tests/golden/cox_sync.npz).  usage: host_tables_check <mod> <rate> <out.f32>
// layout: tI[1152] tQ[1152] energy_ref[1] preamble[8064]
#include <cstdio>
#include <cstdlib>
#include "../../ria_amd/csrc/host_tables.hpp"
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const ria::CarrierPlan p = ria::build_carrier_plan(atoi(argv[1]), atoi(argv[2]));
    const ria::CoxTemplate t = ria::build_cox_template(p);
    const std::vector<float> pre = ria::build_cox_preamble(p);
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 3;
    fwrite(t.tI.data(), 4, t.tI.size(), f);
    fwrite(t.tQ.data(), 4, t.tQ.size(), f);
    fwrite(&t.energy_ref, 4, 1, f);
    fwrite(pre.data(), 4, pre.size(), f);
    fclose(f);
    printf("%zu %zu\n", t.tI.size(), pre.size());
    return 0;
}
