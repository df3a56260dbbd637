/*
 * featMatchMultiple.c -- the reference's matcher command line over the C-ABI.
 *
 * Command line, file names and formats follow R/featMatchMultiple/featMatchMultiple.cpp:405-646 (R/ =
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): options -o <report>, -s<0|1|2>, -r / -r-, -n <neighbours>,
 * -f <list file>; side files _command.txt, _names.txt, feature_count.txt.  What it computes is the reference's all-to-all
 * matching (matchAllToAll, :18-146): every feature of every image asks for its nearest neighbours among all features, and
 * the hits in other images become soft votes between images -- matching_votes.txt and vote_count.txt.  Two deliberate
 * differences, both forced by the reference as it stands:
 *   - the search is exact (sift3d_knn64 on the GPU) where the reference asks FLANN's randomised kd-tree forest, which is
 *     neither deterministic nor part of /root/reference;
 *   - the reference's main() calls matchAllToOne (:640), whose nearest-neighbour step has its distance computation commented
 *     out (featMatchUtilities.cpp:348-362: every distance is the constant 0), so that path cannot be restated as a working
 *     program; the all-to-all path is the one that still computes what its name says.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "keyfile.h"
#include "match.h"
#include "sift3d.h"

static void usage(void)
{
    printf("Volumetric Feature matching v1.1\n");
    printf("Determines robust alignment solution mapping coordinates in image 2, 3, ... to image 1.\n");
    printf("Usage: %s [options] <input keys 1> <input keys 2> ... \n", "featMatchMultiple");
    printf("  <input keys 1, ...>: input key files, produced from featExtract.\n");
    printf("  <output transform>: output text file with linear transform from keys 2 -> keys 1.\n");
}

typedef struct {
    sift3d_feature *f;
    int64_t n;
} key_set;

/* one all-to-all pass over the given sets (already filtered) */
static int match_all(char **names, key_set *sets, int n_sets, int neighbours, const char *title, int append, int device)
{
    (void)names;
    int rc = -1;
    int64_t *first = (int64_t *)calloc((size_t)n_sets + 1, sizeof(int64_t));
    int32_t *labels = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_sets);
    for (int i = 0; i < n_sets; i++) {
        first[i + 1] = first[i] + sets[i].n;
        labels[i] = i; /* no label file on this command line: an image is its own label (:541-548) */
    }
    const int64_t total = first[n_sets];
    printf("Creating NN index structure, NN=%d, image split=%d, type features=%s\n", neighbours, -1, title);
    printf("Descriptor size: %d\n", SIFT3D_DESC_LEN);
    sift3d_feature *all = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (size_t)(total > 0 ? total : 1));
    int8_t *desc = (int8_t *)malloc((size_t)(total > 0 ? total : 1) * SIFT3D_DESC_LEN);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1) * (size_t)neighbours);
    int32_t *d2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1) * (size_t)neighbours);
    float *votes = (float *)calloc((size_t)n_sets * (size_t)n_sets, sizeof(float));
    int32_t *counts = (int32_t *)calloc((size_t)n_sets * (size_t)n_sets, sizeof(int32_t));
    char err[256] = "";
    if (!all || !desc || !idx || !d2 || !votes || !counts) goto done;
    for (int i = 0; i < n_sets; i++)
        if (sets[i].n) memcpy(all + first[i], sets[i].f, sizeof(sift3d_feature) * (size_t)sets[i].n);
    if (sift3d_match_descriptors(all, total, desc) != 0) {
        printf("Error: a descriptor value is outside 0..127 (not a rank descriptor)\n");
        goto done;
    }
    printf("done.\n");
    if (total > 0) {
        if (sift3d_knn64(device, desc, total, desc, total, neighbours, idx, d2, 1, NULL, err, sizeof err) != SIFT3D_OK) {
            printf("Error: nearest-neighbour search failed: %s\n", err);
            goto done;
        }
        for (int i = 0; i < n_sets; i++) printf("Searching image %d of %d ... \n", i, n_sets);
        if (sift3d_match_votes(all, first, n_sets, labels, n_sets, idx, d2, neighbours, votes, counts) != 0) goto done;
    }
    if (sift3d_match_write_votes("matching_votes.txt", "vote_count.txt", title, votes, counts, n_sets, n_sets, append) != 0) {
        printf("Error: could not write matching_votes.txt / vote_count.txt\n");
        goto done;
    }
    rc = 0;
done:
    free(first); free(labels); free(all); free(desc); free(idx); free(d2); free(votes); free(counts);
    return rc;
}

int main(int argc, char **argv)
{
    if (sift3d_abi_version() != SIFT3D_ABI_VERSION) { /* the library writes whole structures through this program's pointers */
        fprintf(stderr, "featMatchMultiple: libsift3d_hip.so has ABI version %d, this program was built against %d\n", sift3d_abi_version(), SIFT3D_ABI_VERSION);
        return -1;
    }
    if (argc < 3) {
        usage();
        return -1;
    }
    FILE *cf = fopen("_command.txt", "wt");
    if (cf) {
        for (int i = 0; i < argc; ++i) fprintf(cf, "%s ", argv[i]);
        fprintf(cf, "\n");
        fclose(cf);
    }
    int a = 1, only_reoriented = 1, peaks_mode = 4, neighbours = 5;
    const char *report = "report.txt", *list_file = NULL;
    while (a < argc && argv[a][0] == '-') {
        switch (argv[a][1]) {
        case 'o': case 'O':
            a++;
            if (a >= argc) { usage(); return -1; }
            report = argv[a++];
            break;
        case 's': case 'S':
            peaks_mode = atoi(&argv[a][2]);
            a++;
            break;
        case 'r': case 'R':
            only_reoriented = argv[a][2] == '-' ? 0 : 1;
            a++;
            break;
        case 'n': case 'N':
            a++;
            if (a >= argc) { usage(); return -1; }
            neighbours = atoi(argv[a++]);
            break;
        case 'f': case 'F':
            a++;
            if (a >= argc) { usage(); return -1; }
            list_file = argv[a++];
            break;
        default:
            printf("Error: unknown command line argument: %s\n", argv[a]);
            return -1;
        }
    }
    if (neighbours < 1 || neighbours > 32) {
        printf("Error: the number of neighbours must be 1..32\n");
        return -1;
    }
    FILE *rf = fopen(report, "wt");
    if (rf) fclose(rf);
    /* names: from the list file (one per non-empty line) or the rest of the command line */
    char **names = NULL;
    int n_names = 0;
    if (list_file) {
        FILE *lf = fopen(list_file, "rt");
        if (!lf) {
            printf("Error: could not read input file name list: %s\n", list_file);
            return -1;
        }
        char line[4096];
        while (fgets(line, sizeof line, lf)) {
            size_t len = strlen(line);
            while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
            if (!len) continue;
            names = (char **)realloc(names, sizeof(char *) * (size_t)(n_names + 1));
            names[n_names++] = strdup(line);
        }
        fclose(lf);
    } else {
        for (int i = a; i < argc; i++) {
            names = (char **)realloc(names, sizeof(char *) * (size_t)(n_names + 1));
            names[n_names++] = argv[i];
        }
    }
    if (n_names < 1) {
        usage();
        return -1;
    }
    FILE *nf = fopen("_names.txt", "wt");
    if (nf) {
        for (int i = 0; i < n_names; i++) fprintf(nf, "%s\t%d\n", names[i], i);
        fclose(nf);
    }
    key_set *sets = (key_set *)calloc((size_t)n_names, sizeof(key_set)), *peaks = NULL, *valleys = NULL;
    const char *title = "Peak and Valley";
    if (peaks_mode == 2) {
        peaks = (key_set *)calloc((size_t)n_names, sizeof(key_set));
        valleys = (key_set *)calloc((size_t)n_names, sizeof(key_set));
    }
    int64_t total = 0;
    /* every name keeps its set and its index: a file that cannot be read stays an empty set (the reference's loop leaves
     * iFeatVec == iFeatVecTotal, featMatchMultiple.cpp:578-632), so feature_count.txt and the vote matrices have one row
     * per line of _names.txt whatever opened */
    const int n_read = n_names;
    for (int i = 0; i < n_names; i++) {
        const char *pch = strrchr(names[i], '\\');
        pch = pch ? pch + 1 : names[i];
        printf("Reading file %d: %s...", i, pch);
        sift3d_feature *f = NULL;
        int64_t n = 0;
        if (sift3d_read_key(names[i], &f, &n) != 0) {
            printf("Error: could not open feature file %d: %s\n", i, names[i]);
            continue; /* as the reference does: the set stays empty, later files keep their index */
        }
        n = sift3d_match_filter(f, n, only_reoriented, peaks_mode == 0 ? 0 : (peaks_mode == 1 ? 1 : 4));
        if (peaks_mode == 0) title = "Peaks";
        if (peaks_mode == 1) title = "Valley";
        if (peaks_mode == 2) { /* SplitFeatures (featMatchUtilities.cpp:1342-1370): set 0 without the flagged records, set 1 with only them */
            peaks[i].f = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (size_t)(n ? n : 1));
            valleys[i].f = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (size_t)(n ? n : 1));
            memcpy(peaks[i].f, f, sizeof(sift3d_feature) * (size_t)n);
            memcpy(valleys[i].f, f, sizeof(sift3d_feature) * (size_t)n);
            peaks[i].n = sift3d_match_filter(peaks[i].f, n, only_reoriented, 0);
            valleys[i].n = sift3d_match_filter(valleys[i].f, n, only_reoriented, 1);
        }
        sets[i].f = f;
        sets[i].n = n;
        total += n;
        printf("feats: %d, total: %d\n", (int)n, (int)total);
    }
    FILE *fc = fopen("feature_count.txt", "wt");
    if (fc) {
        for (int i = 0; i < n_read; i++) fprintf(fc, "%d\t%d\n", i, (int)sets[i].n);
        fclose(fc);
    }
    if (sift3d_device_count() <= 0) {
        printf("Error: no HIP device (there is no CPU path in this build)\n");
        return -1;
    }
    /* -s2: all three passes append to matching_votes.txt / vote_count.txt (featMatchMultiple.cpp:58-65: "at" whenever
     * bOnlyPeaksFeatures == 2, the first pass included) */
    int rc = match_all(names, sets, n_read, neighbours, title, peaks_mode == 2, 0);
    if (rc == 0 && peaks_mode == 2) {
        rc = match_all(names, peaks, n_read, neighbours, "Peaks", 1, 0);
        if (rc == 0) rc = match_all(names, valleys, n_read, neighbours, "Valley", 1, 0);
    }
    return rc == 0 ? 0 : -1;
}
