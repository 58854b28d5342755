/* The C ABI from plain C: include/jenga_amd.h must be a valid C99 header and the library must link and answer the
 * two entry points that need no GPU.  Built and run by tests/test_capi_cpu.py. */
#include <stdio.h>
#include <string.h>

#include "jenga_amd.h"

int main(void) {
    int v = jenga_abi_version();
    const char* e = jenga_last_error();
    if (v != JENGA_ABI_VERSION) {
        printf("abi version %d != header %d\n", v, JENGA_ABI_VERSION);
        return 1;
    }
    if (e == NULL) {
        printf("jenga_last_error returned NULL\n");
        return 2;
    }
    /* argument validation happens before any device work: a null pointer is rejected with a message */
    if (jenga_gather_rows(NULL, NULL, NULL, NULL, 1, 1, 16, 16, 16) == 0) {
        printf("jenga_gather_rows accepted null pointers\n");
        return 3;
    }
    if (strlen(jenga_last_error()) == 0) {
        printf("no error message after a rejected call\n");
        return 4;
    }
    printf("ok abi=%d block=%d head_dim=%d\n", v, JENGA_BLOCK, JENGA_HEAD_DIM);
    return 0;
}
