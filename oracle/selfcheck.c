/* selfcheck.c — small driver exercising every oracle entry point; built with
 * -fsanitize=address,undefined by `make -C oracle selfcheck` (sanitizers run on the CPU build only). */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void) {
    const uint32_t d = 48, n = 1500, nq = 40;
    float *X = malloc((size_t)n * d * 4), *Q = malloc((size_t)nq * d * 4);
    orc_gen_rows(7, d, 16, 32, 1.0f, 0, 0, n, X);
    orc_gen_rows(7, d, 16, 32, 1.0f, 1, 0, nq, Q);
    orc_graph *g = orc_hnsw_build(X, n, d, 8, 32, 3);
    uint64_t keys[2][40 * 10]; float dist[2][40 * 10]; uint32_t cnt[2][40]; uint64_t st[40 * 3];
    orc_graph_search_batch(g, Q, nq, 10, 50, 0, 4, keys[0], dist[0], cnt[0], st);
    orc_graph_search_batch(g, Q, nq, 10, 50, 1, 2, keys[1], dist[1], cnt[1], NULL);
    if (memcmp(keys[0], keys[1], sizeof keys[0]) || memcmp(dist[0], dist[1], sizeof dist[0])) { puts("FAIL: algo 0 != algo 1"); return 1; }
    uint64_t info[8]; orc_graph_info(g, info);
    uint8_t *lv = malloc(n); uint32_t *uo = malloc(n * 4), *a0 = malloc((size_t)n * info[4] * 4), *aU = malloc((info[7] + 1) * info[3] * 4);
    orc_graph_export(g, lv, uo, a0, aU);
    orc_graph *g2 = orc_graph_from_arrays(X, n, d, d, (uint32_t)info[3], (uint32_t)info[4], (uint32_t)info[5], (uint32_t)info[6], lv, uo, a0, aU, info[7]);
    uint64_t k2[10]; float d2[10]; uint32_t c2; uint64_t s2[3];
    orc_graph_search(g2, Q, 10, 50, 0, k2, d2, &c2, s2);
    if (memcmp(k2, keys[0], sizeof k2)) { puts("FAIL: export/import"); return 1; }
    orc_graph *v = orc_vamana_build(X, 600, d, 12, 24, 1.2f, 5);
    orc_graph_search(v, Q, 10, 30, 1, k2, d2, &c2, s2);
    uint64_t sk[5]; float ss[5]; uint32_t sn; uint8_t mask[(1500 + 7) / 8]; memset(mask, 0x55, sizeof mask);
    orc_scan_topk(X, n, d, Q, 5, 0, mask, sk, ss, &sn);
    uint64_t mk[10]; float md[10]; uint32_t mn;
    orc_merge_topk(keys[0], dist[0], cnt[0], 4, 10, 10, mk, md, &mn);
    uint64_t hi[3] = {0, 1, 2}, ho[3]; float hv[3] = {0.9f, 0.8f, 0.7f}, hb[3] = {0.5f, 0.9f, 0.3f}, hs[3];
    orc_hybrid_rerank(hi, hv, 3, hb, 3, 0.5f, ho, hs);
    uint16_t *F = malloc(100 * 32 * 2), *W = malloc(32 * d * 2); float *E = malloc(100 * d * 4);
    orc_synth_features(1, 32, 8, 8, 1.0f, 0, 0, 100, F); orc_synth_weights(1, 32, d, W); orc_recompute_encode(F, 100, 32, W, d, E);
    orc_graph_free(g); orc_graph_free(g2); orc_graph_free(v);
    free(X); free(Q); free(lv); free(uo); free(a0); free(aU); free(F); free(W); free(E);
    puts("oracle selfcheck OK");
    return 0;
}
