// Host-side check of ceg_pairfrac::build_erfc_table (tests/test_host_logic.py): writes base, ni, worst and the records to stdout.
#include "../../crystalenergygrids.jl_amd/csrc/ceg_pairfrac.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    ceg_pairfrac::ErfcTable et;
    const bool ok = ceg_pairfrac::build_erfc_table(atof(argv[1]), atof(argv[2]), atof(argv[3]), et);
    printf("%d %d %d %.17g %d\n", (int)ok, et.base, et.ni, et.worst, ceg_pairfrac::ERFC_REC);
    for (double x : et.rec) printf("%.17g\n", x);
    return 0;
}
