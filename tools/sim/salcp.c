// experiment tool: SA (SA-IS of the oracle) + LCP (Kasai) of one block -> binary files
#include "../../oracle/bwt.c"
#include <stdio.h>
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* src = malloc(n); fread(src, 1, n, f); fclose(f);
    int* s = malloc(sizeof(int) * (n + 1)); int* SA = malloc(sizeof(int) * (n + 1));
    for (long i = 0; i < n; i++) s[i] = src[i] + 1; s[n] = 0;
    sa_is(s, SA, n + 1, 256);
    int* sa = SA + 1;
    int* rank = s;   // reuse
    for (long i = 0; i < n; i++) rank[sa[i]] = i;
    int* lcp = malloc(sizeof(int) * n);
    long h = 0; lcp[0] = 0;
    for (long i = 0; i < n; i++) {
        if (rank[i] > 0) { long j = sa[rank[i] - 1]; while (i + h < n && j + h < n && src[i + h] == src[j + h]) h++; lcp[rank[i]] = h; if (h > 0) h--; }
        else h = 0;
    }
    f = fopen(argv[2], "wb"); fwrite(sa, 4, n, f); fclose(f);
    f = fopen(argv[3], "wb"); fwrite(lcp, 4, n, f); fclose(f);
    return 0;
}
