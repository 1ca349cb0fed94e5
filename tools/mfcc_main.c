/* `mfcc <config.ini>` -- command-line front end with the reference's usage (README.md:26,
 * main.c:98-300): reads the ini, converts every "src<TAB>dst" line of its file list. */
#include <stdio.h>

#include "mfcc_amd.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s config.ini\n", argv[0]);
        return 2;
    }
    const int rc = MFCCWapperList(argv[1]);
    if (rc < 0) {
        fprintf(stderr, "mfcc: %s\n", MFCCLastError());
        return 1;
    }
    if (rc > 0) {
        fprintf(stderr, "mfcc: %s\n", MFCCLastError());
        return 1;
    }
    return 0;
}
