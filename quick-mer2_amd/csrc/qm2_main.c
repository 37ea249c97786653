/*
 * qm2_main.c — command dispatcher (QuicKmer.c:1485-1519): `count` (the accelerated
 * path) and its consumer `est` (host C, qm2_est.c).  `index`, `search` and `sparse`
 * stay with the reference binary (`qm2-mkdict` covers `search -e 0`); the on-disk
 * formats between them are unchanged.
 */
#include <stdio.h>
#include <string.h>

#include "qm2_host.h"

static void usage(void)
{
    puts("quicKmer2 (MI355X count path)\n");
    puts("Operation modes:");
    puts("\tcount\tCK-mer counting from library input (this build: GPU)");
    puts("\test\tCopy-number estimation from count's output (host)");
    puts("\tindex, search, sparse\tuse the reference quicKmer2 (or qm2-mkdict for search -e 0); file formats are identical");
}

int main(int argc, char **argv)
{
    if (argc < 2) { usage(); return 1; }
    if (strcmp(argv[1], "count") == 0) return qm2_main_count(argc - 1, argv + 1);   /* :1501-1502 */
    if (strcmp(argv[1], "est") == 0) return qm2_main_est(argc - 1, argv + 1);       /* :1505 */
    if (!strcmp(argv[1], "index") || !strcmp(argv[1], "search") || !strcmp(argv[1], "sparse")) {
        printf("'%s' is not part of the accelerated path: run it with the reference quicKmer2 "
               "(.qm/.qgc/.bed/.bin/.txt are format-identical)\n", argv[1]);
        return 1;
    }
    usage();
    return 1;
}
