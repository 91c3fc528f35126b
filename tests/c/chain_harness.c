/* chain_harness — runs papr_exact_chain (host-side product code, dtv-utils_amd/csrc/papr_host.c) on
 * program files given on the command line, for the AddressSanitizer / UBSan run in
 * tests/test_sanitizers.py.  Prints "<return code> <sum as hex double>" per invocation. */
#include <stdio.h>
#include <stdlib.h>

#include "papr_hip.h"

int main(int argc, char **argv)
{
    const void *progs[16];
    size_t sizes[16];
    int n = 0;
    for (int a = 1; a < argc && n < 16; a++) {
        FILE *fp = fopen(argv[a], "rb");
        if (!fp)
            return 2;
        fseek(fp, 0, SEEK_END);
        long len = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        /* exact-size heap block: any read past the program's end is an ASan report */
        unsigned char *buf = (unsigned char *)malloc(len > 0 ? (size_t)len : 1);
        if (len > 0 && fread(buf, 1, (size_t)len, fp) != (size_t)len)
            return 2;
        fclose(fp);
        progs[n] = buf;
        sizes[n] = (size_t)len;
        n++;
    }
    double sum = 0.0;
    int rc = papr_exact_chain(progs, sizes, n, &sum);
    printf("%d %a\n", rc, sum);
    for (int k = 0; k < n; k++)
        free((void *)progs[k]);
    return 0;
}
