// pool_harness — dtv-utils_amd/csrc/ts_line_pool.h on its own (built with -fsanitize=thread by tests/test_sanitizers.py):
//   pool_harness <threads> <rounds>   prints the self-test's verdict (0: every job of every round ran exactly once)
#include <cstdio>
#include <cstdlib>

#include "ts_line_pool.h"

int main(int argc, char **argv)
{
    if (argc != 3)
        return 2;
    printf("%d\n", ts_line_pool_selftest(atoi(argv[1]), atoi(argv[2])));
    return 0;
}
