/*
 * ghmm_rendezvous.c — the id exchange behind ghmm_comm_create_file, host code only.
 *
 * The reference has no communication of any kind (single-threaded C, TF:238-358); what the
 * ranks share is the accumulator sum of SURVEY.md §8(e) (TF:1614, 1618, 1660, 1716-1722,
 * 318-320), and an RCCL communicator for that sum needs one 128-byte id carried from rank 0 to
 * every other rank.  The C trainer's rank mode (train_main.c, GHMM_COMM_ID) carries it through
 * a file; this file is that protocol, separable from ncclCommInitRank so that it can be driven
 * by plain CPU processes (tests/test_host.py).
 *
 * Protocol (all files written to a temporary name and renamed: a reader sees a whole file or
 * none):
 *   rank r > 0   writes  <path>.join.<r>   = { magic, r, nonce_r }      (nonce: /dev/urandom)
 *                polls   <path>            until it holds nonce_r at slot r, takes the id,
 *                removes <path>.join.<r>   (= "joined")
 *   rank 0       polls   <path>.join.<r>   for every r, writes <path> = { magic, world,
 *                nonce_1..nonce_{world-1}, id }, writes it AGAIN whenever a join file shows a
 *                nonce other than the one published (a stale join file of a crashed job that
 *                its rank has since replaced), waits until every join file has gone, removes
 *                <path>.
 * A stale <path> of an earlier job cannot be taken for the new id: it does not hold the
 * fresh nonce.  Every wait is bounded by timeout_s (CLOCK_MONOTONIC) and ends in GHMM_ERR_IO.
 */
#define _GNU_SOURCE
#include "ghmm.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

void ghmm_set_error(const char *fmt, ...);

#define RDV_MAGIC 0x31305644524d4847ull /* "GHMRDV01" */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static uint64_t fresh_nonce(int rank)
{
    uint64_t v = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) {
        if (fread(&v, sizeof v, 1, f) != 1) v = 0;
        fclose(f);
    }
    if (!v) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        v = ((uint64_t)ts.tv_nsec << 20) ^ ((uint64_t)ts.tv_sec << 1) ^ ((uint64_t)getpid() << 40) ^
            (uint64_t)rank ^ 0x9E3779B97F4A7C15ull;
    }
    return v ? v : 1;
}

/* whole file or nothing: write beside the target, then rename over it */
static int write_atomic(const char *path, const void *buf, size_t n)
{
    size_t len = strlen(path) + 48;
    char *tmp = (char *)malloc(len);
    if (!tmp) return -1;
    snprintf(tmp, len, "%s.tmp.%ld", path, (long)getpid());
    FILE *f = fopen(tmp, "wb");
    int ok = f && fwrite(buf, 1, n, f) == n;
    if (f && fclose(f) != 0) ok = 0;
    if (ok && rename(tmp, path) != 0) ok = 0;
    if (!ok) (void)unlink(tmp);
    free(tmp);
    return ok ? 0 : -1;
}

/* 1 = read n bytes, 0 = absent / another size (not yet there, or not ours) */
static int read_whole(const char *path, void *buf, size_t n)
{
    struct stat st;
    if (stat(path, &st) != 0 || st.st_size != (off_t)n) return 0;
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    size_t got = fread(buf, 1, n, f);
    fclose(f);
    return got == n;
}

int ghmm_rendezvous_file(const char *path, int rank, int world, double timeout_s, void *id_bytes)
{
    if (!path || !*path || !id_bytes || world < 1 || rank < 0 || rank >= world || !(timeout_s >= 0.0)) {
        ghmm_set_error("ghmm_rendezvous_file: bad arguments");
        return GHMM_ERR_ARG;
    }
    if (world == 1) return GHMM_OK;
    const size_t plen = strlen(path) + 32;
    char *jp = (char *)malloc(plen);
    const size_t rec_n = (size_t)(2 + (world - 1)) * 8 + GHMM_COMM_ID_BYTES;
    unsigned char *rec = (unsigned char *)malloc(rec_n);
    uint64_t *seen = (uint64_t *)calloc((size_t)world, sizeof(uint64_t));   /* nonce read from join.<r> */
    uint64_t *pub = (uint64_t *)calloc((size_t)world, sizeof(uint64_t));    /* nonce last published */
    unsigned char *gone = (unsigned char *)calloc((size_t)world, 1);
    int rc = GHMM_OK;
    if (!jp || !rec || !seen || !pub || !gone) {
        ghmm_set_error("ghmm_rendezvous_file: out of memory");
        rc = GHMM_ERR_ALLOC;
        goto done;
    }
    const double t_end = now_s() + timeout_s;
    const useconds_t nap = 2000;

    if (rank > 0) {
        uint64_t join[3] = {RDV_MAGIC, (uint64_t)rank, fresh_nonce(rank)};
        snprintf(jp, plen, "%s.join.%d", path, rank);
        if (write_atomic(jp, join, sizeof join) != 0) {
            ghmm_set_error("rank %d: cannot write %s", rank, jp);
            rc = GHMM_ERR_IO;
            goto done;
        }
        for (;;) {
            if (read_whole(path, rec, rec_n)) {
                uint64_t head[2], mine;
                memcpy(head, rec, sizeof head);
                memcpy(&mine, rec + 8 * (size_t)(1 + rank), 8);
                if (head[0] == RDV_MAGIC && head[1] == (uint64_t)world && mine == join[2]) {
                    memcpy(id_bytes, rec + rec_n - GHMM_COMM_ID_BYTES, GHMM_COMM_ID_BYTES);
                    break;
                }
            }
            if (now_s() >= t_end) {
                (void)unlink(jp);
                ghmm_set_error("rank %d: no communicator id for this job in %s after %.1f s", rank, path,
                               timeout_s);
                rc = GHMM_ERR_IO;
                goto done;
            }
            usleep(nap);
        }
        (void)unlink(jp); /* joined */
        goto done;
    }

    /* rank 0 */
    {
        int published = 0;
        for (;;) {
            int all_seen = 1, all_gone = 1, changed = 0;
            for (int r = 1; r < world; r++) {
                if (gone[r]) continue;
                uint64_t join[3];
                snprintf(jp, plen, "%s.join.%d", path, r);
                if (read_whole(jp, join, sizeof join) && join[0] == RDV_MAGIC && join[1] == (uint64_t)r) {
                    seen[r] = join[2];
                    if (seen[r] != pub[r]) changed = 1;
                } else if (published && pub[r] && access(jp, F_OK) != 0) {
                    gone[r] = 1; /* the rank has taken the id and removed its file */
                    continue;
                }
                if (!seen[r]) all_seen = 0;
                all_gone = 0;
            }
            if (all_gone && published) break;
            if (all_seen && (changed || !published)) {
                uint64_t head[2] = {RDV_MAGIC, (uint64_t)world};
                memcpy(rec, head, sizeof head);
                for (int r = 1; r < world; r++) memcpy(rec + 8 * (size_t)(1 + r), &seen[r], 8);
                memcpy(rec + rec_n - GHMM_COMM_ID_BYTES, id_bytes, GHMM_COMM_ID_BYTES);
                if (write_atomic(path, rec, rec_n) != 0) {
                    ghmm_set_error("cannot publish the communicator id in %s", path);
                    rc = GHMM_ERR_IO;
                    goto done;
                }
                memcpy(pub, seen, (size_t)world * sizeof(uint64_t));
                published = 1;
            }
            if (now_s() >= t_end) {
                int miss = 0;
                for (int r = 1; r < world; r++)
                    if (!gone[r]) { miss = r; break; }
                ghmm_set_error("rank 0: rank %d has not %s within %.1f s (%s)", miss,
                               seen[miss] ? "taken the communicator id" : "announced itself", timeout_s, path);
                rc = GHMM_ERR_IO;
                break;
            }
            usleep(nap);
        }
        if (published) (void)unlink(path);
    }
done:
    free(jp);
    free(rec);
    free(seen);
    free(pub);
    free(gone);
    return rc;
}
