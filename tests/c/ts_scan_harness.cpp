// ts_scan_harness — the packet scan through the C ABI, for tools/tsan_cli.sh (built with the host code under ThreadSanitizer,
// run on the GPU box): a damaged synthetic stream scanned `scans` times by one context; every report must be the first one's.
//   ts_scan_harness <npackets> <period> <scans>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ts_hip.h"

int main(int argc, char **argv)
{
    if (argc != 4)
        return 2;
    const unsigned long long period = strtoull(argv[2], nullptr, 0);
    const unsigned long long n = strtoull(argv[1], nullptr, 0) / (4 * period) * (4 * period);
    const int scans = atoi(argv[3]);
    ts_hip_ctx *ctx = nullptr;
    if (ts_hip_open(&ctx, 0) != 0) {
        printf("open failed: %s\n", ts_hip_last_error(nullptr));
        return 1;
    }
    if (ts_hip_generate_damaged(ctx, 0x7500001ull, n, period) != 0) {
        printf("generate failed: %s\n", ts_hip_last_error(ctx));
        return 1;
    }
    std::vector<char> res_mem(ts_hip_result_size());
    ts_scan_result *res = reinterpret_cast<ts_scan_result *>(res_mem.data());
    std::string first;
    int bad = 0;
    for (int k = 0; k < scans; k++) {
        if (ts_hip_scan(ctx, 0, res) != 0) {
            printf("scan failed: %s\n", ts_hip_last_error(ctx));
            return 1;
        }
        const unsigned long long ne = ts_hip_sync_error_count(ctx), nd = ts_hip_discontinuity_count(ctx);
        std::vector<ts_sync_error> errs(ne ? ne : 1);
        std::vector<ts_discontinuity> discs(nd ? nd : 1);
        ts_hip_get_sync_errors(ctx, 0, ne, errs.data());
        ts_hip_get_discontinuities(ctx, 0, nd, discs.data());
        std::string rep((size_t)(64 * (ne + nd) + (1 << 20)), '\0');
        const size_t len = ts_format_report_all(res, errs.data(), ne, discs.data(), nd, &rep[0], rep.size());
        rep.resize(len);
        if (k == 0)
            first = rep;
        else if (rep != first)
            bad++;
    }
    printf("%d scans of %llu packets, one damaged spot per %llu: %zu bytes of report, %d differ from the first\n", scans, n, period, first.size(), bad);
    ts_hip_close(ctx);
    return bad != 0;
}
