/*
 * examples/commit.c -- a plain C caller of the drop-in library: the same calls a program written
 * against the reference's ckzg.h makes (load_trusted_setup_file, blob_to_kzg_commitment,
 * compute_blob_kzg_proof, verify_blob_kzg_proof, free_trusted_setup), compiled with a C compiler
 * and linked against libckzg_hip.so instead of ckzg.c + libblst.
 *
 *   gcc -std=c11 -Iinclude examples/commit.c -Lc-kzg-4844_amd -lckzg_hip \
 *       -Wl,-rpath,$PWD/c-kzg-4844_amd -o examples/commit
 *   ./examples/commit c-kzg-4844_amd/data/trusted_setup.txt
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ckzg.h"

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "c-kzg-4844_amd/data/trusted_setup.txt";
    FILE *fp = fopen(path, "r");
    if (!fp) {
        fprintf(stderr, "cannot open %s\n", path);
        return 2;
    }
    KZGSettings s;
    C_KZG_RET ret = load_trusted_setup_file(&s, fp, 0);
    fclose(fp);
    if (ret != C_KZG_OK) {
        fprintf(stderr, "load_trusted_setup_file failed: %d (no GPU?)\n", (int)ret);
        return 3;
    }
    Blob *blob = calloc(1, sizeof(Blob));
    for (size_t i = 0; i < FIELD_ELEMENTS_PER_BLOB; i++) {
        blob->bytes[32 * i + 31] = (uint8_t)(i & 0xff);
        blob->bytes[32 * i + 30] = (uint8_t)(i >> 8);
    }
    KZGCommitment c;
    KZGProof p;
    bool ok = false;
    ret = blob_to_kzg_commitment(&c, blob, &s);
    if (ret == C_KZG_OK) ret = compute_blob_kzg_proof(&p, blob, &c, &s);
    if (ret == C_KZG_OK) ret = verify_blob_kzg_proof(&ok, blob, &c, &p, &s);
    printf("commitment ");
    for (int i = 0; i < 48; i++) printf("%02x", c.bytes[i]);
    printf("\nret=%d verified=%d\n", (int)ret, (int)ok);
    free(blob);
    free_trusted_setup(&s);
    return (ret == C_KZG_OK && ok) ? 0 : 1;
}
