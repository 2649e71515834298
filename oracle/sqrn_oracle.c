/*
 * oracle/sqrn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded, fp64 restatement of the single-sequence folding
 * core of febos/SQUARNA v3.2.2 (reference: src/SQUARNA/SQRNdbnseq.py and
 * SQRNalgos.py).  It keeps the reference's algorithmic form -- a dense N x N
 * matrix, a full re-scan of every anti-diagonal per AnnotateStems call, a
 * per-position walk per candidate stem -- so that it is an independent check
 * of the MI355X path (which uses a different decomposition) and a
 * representative CPU baseline.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (squarna_amd/) never links or calls it.
 *
 * Parity is pinned by tests/test_oracle_golden.py against vectors generated
 * from the reference itself (tests/golden/gen_golden.py).
 *
 * Known deviation: bpp != 0 (ViennaRNA) is not restated -- parity unpinned
 * for that branch (SQRNdbnseq.py:341-364); the oracle rejects it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static int is_sep(char c) { return c == ';' || c == '&'; }   /* SQRNdbnseq.py:14 */

/* --------------------------------------------------------------------------
 * BPMatrix -- SQRNdbnseq.py:258-367 (bpp_power == 0 branch only).
 * flags[i]: bit0 = i in rxs, bit1 = i in rlefts, bit2 = i in rrights
 * (ParseRestraints, SQRNdbnseq.py:370-376).  wkeys = 2*nw chars ("GCAUGU").
 * -------------------------------------------------------------------------- */
ORC_API int orc_bpmatrix(const char *seq, int n, const char *wkeys, const double *wvals, int nw,
                         const uint8_t *flags, int interchainonly, const double *reacts,
                         double *boolmat, double *scoremat)
{
    static double wtab[256][256];
    static unsigned char inbps[256][256];
    memset(wtab, 0, sizeof wtab);
    memset(inbps, 0, sizeof inbps);
    for (int k = 0; k < nw; k++) {                       /* :282-284 */
        unsigned char a = (unsigned char)wkeys[2 * k], b = (unsigned char)wkeys[2 * k + 1];
        wtab[a][b] = wvals[k]; inbps[a][b] = 1;
        wtab[b][a] = wvals[k]; inbps[b][a] = 1;
    }
    int *chains = (int *)calloc((size_t)n + 1, sizeof(int)); /* :264-271 */
    if (interchainonly) {
        int curr = 0;
        for (int i = 0; i < n; i++) {
            if (is_sep(seq[i])) curr++;
            else chains[i] = curr;
        }
    }
    int defaultreacts = 1;                               /* :273 */
    if (reacts)
        for (int i = 0; i < n; i++)
            if (reacts[i] != 0.5) { defaultreacts = 0; break; }

    memset(boolmat, 0, sizeof(double) * (size_t)n * n);
    memset(scoremat, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n - 1; i++) {
        int inc4 = 4;                                    /* :294-297 */
        for (int chk = 1; chk <= 2; chk++)
            if (i + chk < n && is_sep(seq[i + chk])) inc4 = chk + 1;
        for (int j = i + inc4; j < n; j++) {
            unsigned char a = (unsigned char)seq[i], b = (unsigned char)seq[j];
            double bl = (double)inbps[a][b];             /* :300-304 */
            bl *= (double)(!interchainonly || chains[i] != chains[j]);
            bl *= (double)(!(flags[i] & 1) && !(flags[j] & 1));
            bl *= (double)(!(flags[j] & 2));
            bl *= (double)(!(flags[i] & 4));
            boolmat[(size_t)i * n + j] = bl;
            double rf;                                   /* :329-336 */
            if (defaultreacts) rf = 1.0;
            else rf = pow((1.0 - (reacts[i] + reacts[j]) / 2.0) * 2.0, 0.5);
            double w = wtab[a][b];
            if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
            scoremat[(size_t)i * n + j] = w * bl * rf;   /* :338 */
        }
    }
    free(chains);
    return 0;
}

/* --------------------------------------------------------------------------
 * stems are (i_outer, j_outer, len) triples; bps are (i+k, j-k), k < len.
 * -------------------------------------------------------------------------- */
typedef struct {
    int i, j, len;
    double bpscore;   /* stem[2] */
    double finscore;  /* stem[3] */
} orc_stem;

typedef struct { orc_stem *v; int n, cap; } stemvec;

static void sv_push(stemvec *s, orc_stem x)
{
    if (s->n == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 64;
        s->v = (orc_stem *)realloc(s->v, sizeof(orc_stem) * (size_t)s->cap);
    }
    s->v[s->n++] = x;
}

/* AnnotateStems -- SQRNdbnseq.py:427-495 with PreStemsFromDiag :379-402 and
 * StemsFromPreStemsDiffEdge :405-418 (diff = 0, span = -1: the only values any
 * caller passes).  rbps = nrbps (v,w) pairs; rstems = triples. */
static void annotate(const double *boolmat, const double *scoremat, int n,
                     const int *rbps, int nrbps, const orc_stem *rstems, int nrstems,
                     double minlen, double minscore, double *work, stemvec *out)
{
    double *m = work;                                    /* :431 matrix copy */
    memcpy(m, boolmat, sizeof(double) * (size_t)n * n);
    for (int k = 0; k < nrbps; k++) {                    /* :438-443 */
        int v = rbps[2 * k], w = rbps[2 * k + 1];
        for (int t = 0; t < n; t++) {
            m[(size_t)v * n + t] = 0; m[(size_t)t * n + v] = 0;
            m[(size_t)w * n + t] = 0; m[(size_t)t * n + w] = 0;
        }
        m[(size_t)v * n + w] = boolmat[(size_t)v * n + w];
    }
    for (int k = 0; k < nrstems; k++)                    /* :446-451 */
        for (int b = 0; b < rstems[k].len; b++) {
            int v = rstems[k].i + b, w = rstems[k].j - b;
            for (int t = 0; t < n; t++) {
                m[(size_t)v * n + t] = 0; m[(size_t)t * n + v] = 0;
                m[(size_t)w * n + t] = 0; m[(size_t)t * n + w] = 0;
            }
        }
    /* :456-457 diagonal starts (0,x) x in [4,N) then (y,N-1) y in [1,N-4) */
    int nd = 0;
    for (int pass = 0; pass < 2; pass++) {
        int lo = pass == 0 ? 4 : 1, hi = pass == 0 ? n : n - 4;
        for (int t = lo; t < hi; t++) {
            int x = pass == 0 ? 0 : t, y = pass == 0 ? t : n - 1;
            int i = x, j = y;
            int first = -1; double sum = 0; int fi = 0, fj = 0;
            nd++;
            while (i <= j - 1) {                         /* :486 */
                double isbp = m[(size_t)i * n + j];
                if (isbp != 0 && first < 0) {            /* :390 */
                    first = i; fi = i; fj = j; sum = 0;
                }
                if (isbp != 0) sum = sum + scoremat[(size_t)i * n + j];  /* :416 left-to-right */
                if (isbp == 0 && first >= 0) {           /* :394 */
                    int len = i - first;
                    if ((double)len >= minlen && sum >= minscore) {      /* :492 */
                        orc_stem s = {fi, fj, len, sum, 0};
                        sv_push(out, s);
                    }
                    first = -1;
                }
                i++; j--;
            }
            if (first >= 0) {                            /* :399 */
                int len = i - first;
                if ((double)len >= minlen && sum >= minscore) {
                    orc_stem s = {fi, fj, len, sum, 0};
                    sv_push(out, s);
                }
            }
        }
    }
    (void)nd;
}

ORC_API int orc_annotate(const double *boolmat, const double *scoremat, int n,
                         const int *rbps, int nrbps, const int *rstems, int nrstems,
                         double minlen, double minscore,
                         int *out_ijl, double *out_score, int cap)
{
    stemvec out = {0};
    orc_stem *rs = (orc_stem *)calloc((size_t)nrstems + 1, sizeof(orc_stem));
    for (int k = 0; k < nrstems; k++) {
        rs[k].i = rstems[3 * k]; rs[k].j = rstems[3 * k + 1]; rs[k].len = rstems[3 * k + 2];
    }
    double *work = (double *)malloc(sizeof(double) * (size_t)n * n + 8);
    annotate(boolmat, scoremat, n, rbps, nrbps, rs, nrstems, minlen, minscore, work, &out);
    int cnt = out.n;
    for (int k = 0; k < cnt && k < cap; k++) {
        out_ijl[3 * k] = out.v[k].i; out_ijl[3 * k + 1] = out.v[k].j; out_ijl[3 * k + 2] = out.v[k].len;
        out_score[k] = out.v[k].bpscore;
    }
    free(work); free(rs); free(out.v);
    return cnt;
}

/* --------------------------------------------------------------------------
 * PairsToDBN(returnlevels / level groups) -- SQRNdbnseq.py:104-163.
 * pairs must be normalised (v<w), sorted and unique.  level[k] = 1-based
 * level of pairs[k] after the size sort; returns the number of groups.
 * -------------------------------------------------------------------------- */
static int crosses(int i, int j, int k, int l)           /* :114-116 */
{
    return (i < k && k < j && j < l) || (k < i && i < l && l < j);
}

ORC_API int orc_pair_levels(const int *pairs, int np, int *level)
{
    if (np == 0) return 0;
    int *cc = (int *)calloc((size_t)np, sizeof(int));
    int *order = (int *)malloc(sizeof(int) * (size_t)np);
    for (int a = 0; a < np; a++) {                       /* :123-124 */
        int c = 0;
        for (int b = 0; b < np; b++)
            if (a != b && crosses(pairs[2 * a], pairs[2 * a + 1], pairs[2 * b], pairs[2 * b + 1])) c++;
        cc[a] = c; order[a] = a;
    }
    /* :125 sorted by (cross_count, p[0]); stable (insertion sort, input sorted) */
    for (int a = 1; a < np; a++) {
        int x = order[a], b = a - 1;
        while (b >= 0 && (cc[order[b]] > cc[x] ||
                          (cc[order[b]] == cc[x] && pairs[2 * order[b]] > pairs[2 * x]))) {
            order[b + 1] = order[b]; b--;
        }
        order[b + 1] = x;
    }
    int *grp = (int *)malloc(sizeof(int) * (size_t)np);    /* group id of each pair */
    int *gsize = (int *)calloc((size_t)np, sizeof(int));
    int ng = 0;
    for (int t = 0; t < np; t++) {                       /* :130-136 first fit */
        int p = order[t], placed = -1;
        for (int g = 0; g < ng && placed < 0; g++) {
            int ok = 1;
            for (int u = 0; u < t && ok; u++) {
                int q = order[u];
                if (grp[q] == g && crosses(pairs[2 * p], pairs[2 * p + 1], pairs[2 * q], pairs[2 * q + 1])) ok = 0;
            }
            if (ok) placed = g;
        }
        if (placed < 0) placed = ng++;
        grp[p] = placed; gsize[placed]++;
    }
    /* :139 groups.sort(key=len, reverse=True) -- stable */
    int *gorder = (int *)malloc(sizeof(int) * (size_t)ng);
    int *grank = (int *)malloc(sizeof(int) * (size_t)ng);
    for (int g = 0; g < ng; g++) gorder[g] = g;
    for (int a = 1; a < ng; a++) {
        int x = gorder[a], b = a - 1;
        while (b >= 0 && gsize[gorder[b]] < gsize[x]) { gorder[b + 1] = gorder[b]; b--; }
        gorder[b + 1] = x;
    }
    for (int r = 0; r < ng; r++) grank[gorder[r]] = r;
    for (int a = 0; a < np; a++) level[a] = grank[grp[a]] + 1;   /* :146-149 */
    free(cc); free(order); free(grp); free(gsize); free(gorder); free(grank);
    return ng;
}

/* --------------------------------------------------------------------------
 * ScoreStems -- SQRNdbnseq.py:607-751 (direct per-position walk).
 * -------------------------------------------------------------------------- */
typedef struct {
    double minlen, minbpscore, minfinscore, bracketweight, distcoef, orderpenalty, loopbonus;
    double suboptmin, suboptmax, suboptsteps, maxstemnum;
} orc_params;

static int goodloop_tab(int a, int b)                    /* :615-622 */
{
    static const int g[19][2] = {{0,0},{0,1},{1,0},{1,1},{0,2},{2,0},{2,2},{1,2},{2,1},{3,1},{1,3},
                                 {2,3},{3,2},{3,3},{3,4},{4,3},{4,4},{4,2},{2,4}};
    for (int k = 0; k < 19; k++) if (g[k][0] == a && g[k][1] == b) return 1;
    return 0;
}

static void score_stems(const char *seq, int n, stemvec *stems, const orc_stem *rstems, int nrstems,
                        const orc_params *p, double minscore)
{
    int *partner = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) partner[i] = -1;          /* :625 */
    int np = 0;
    for (int k = 0; k < nrstems; k++) np += rstems[k].len;
    int *pairs = (int *)malloc(sizeof(int) * 2 * (size_t)(np + 1));
    int *plevel = (int *)malloc(sizeof(int) * (size_t)(np + 1));
    int t = 0;
    for (int k = 0; k < nrstems; k++)                     /* :631-635 */
        for (int b = 0; b < rstems[k].len; b++) {
            int v = rstems[k].i + b, w = rstems[k].j - b;
            partner[v] = w; partner[w] = v;
            pairs[2 * t] = v; pairs[2 * t + 1] = w; t++;
        }
    /* sort pairs (set semantics: stems of one structure never repeat a bp) */
    for (int a = 1; a < np; a++) {
        int x0 = pairs[2 * a], x1 = pairs[2 * a + 1], b = a - 1;
        while (b >= 0 && (pairs[2 * b] > x0 || (pairs[2 * b] == x0 && pairs[2 * b + 1] > x1))) {
            pairs[2 * b + 2] = pairs[2 * b]; pairs[2 * b + 3] = pairs[2 * b + 1]; b--;
        }
        pairs[2 * b + 2] = x0; pairs[2 * b + 3] = x1;
    }
    orc_pair_levels(pairs, np, plevel);                   /* :638 */
    int *levelof = (int *)calloc((size_t)n + 1, sizeof(int)); /* level of the bp a position is in */
    for (int a = 0; a < np; a++) { levelof[pairs[2 * a]] = plevel[a]; levelof[pairs[2 * a + 1]] = plevel[a]; }
    unsigned char *lvseen = (unsigned char *)malloc((size_t)np + 2);

    int keep = 0;
    for (int s = 0; s < stems->n; s++) {                  /* :641 */
        orc_stem *st = &stems->v[s];
        int stemstart = st->i + st->len - 1, stemend = st->j - st->len + 1;   /* :655 bps[-1] */
        int dots = 0, brackets = 0, nlev = 0;
        memset(lvseen, 0, (size_t)np + 2);
        int nblock = 0, be0 = 0, be1 = 0, inblockend = -1, between = 0;
        for (int pos = stemstart + 1; pos < stemend; pos++) {   /* :665-689 */
            int pr = partner[pos];
            if (pr == -1) {
                if (pos > inblockend) dots++;
                if (is_sep(seq[pos])) between = 1;
            } else if (pr < stemstart || pr > stemend) {
                if (pos > inblockend) {
                    brackets++;
                    int lv = levelof[pos];
                    if (!lvseen[lv]) { lvseen[lv] = 1; nlev++; }
                }
            } else if (pos < pr && pr > inblockend) {
                inblockend = pr;
                if (nblock == 0) { be0 = pos; be1 = pr; }
                nblock++;
            }
        }
        int goodloop = 0, diff1 = 0;                      /* :692-698 */
        if (nblock == 1 && goodloop_tab(be0 - stemstart - 1, stemend - be1 - 1)) {
            goodloop = 1;
            diff1 = abs((be0 - stemstart - 1) - (stemend - be1 - 1));
        }
        int goodloopout = 0, diff2 = 0;                   /* :700-711 */
        int os = st->i, oe = st->j;
        int vv = os - 1, ww = oe + 1;
        while (vv >= 0 && os - vv - 1 < 5 && partner[vv] == -1) vv--;
        while (ww < n && ww - oe - 1 < 5 && partner[ww] == -1) ww++;
        {
            int pv = partner[vv >= 0 ? vv : n - 1];       /* :708 python negative index */
            if (pv == ww && ww < n && partner[ww] == vv && goodloop_tab(os - vv - 1, ww - oe - 1)) {
                goodloopout = 1;
                diff2 = abs((os - vv - 1) - (ww - oe - 1));
            }
        }
        double loopfactor = 1 + p->loopbonus * goodloop * (2 - diff1 / 2.0)
                              + p->loopbonus * goodloopout * (2 - diff2 / 2.0);   /* :715 */
        int gnra = 0;                                     /* :598-604,718 */
        if (stemend - stemstart - 1 == 4 && seq[stemstart + 1] == 'G' &&
            (seq[stemstart + 3] == 'G' || seq[stemstart + 3] == 'A') && seq[stemstart + 4] == 'A') gnra = 1;
        double tetrafactor = 1 + 0.25 * gnra;
        double idealdist = inblockend == -1 ? 4 : 2;      /* :721 */
        double stemdist = dots + p->bracketweight * brackets;   /* :723 */
        double sdf = between ? 1.0 : pow(1 / (1 + fabs(stemdist - idealdist)), p->distcoef);   /* :726 */
        double of = pow(1.0 / (1 + nlev), p->orderpenalty);   /* :729 */
        double fin = st->bpscore * sdf * of * loopfactor * 1 * tetrafactor;   /* :732 */
        if (!goodloop && !goodloopout && st->len < 3) fin = -1;   /* :744-745 */
        st->finscore = fin;
        if (fin >= minscore) stems->v[keep++] = *st;      /* :751 */
    }
    stems->n = keep;
    free(partner); free(pairs); free(plevel); free(levelof); free(lvseen);
}

/* ChooseStems -- SQRNdbnseq.py:754-789 */
static int shares_base(const orc_stem *a, const orc_stem *b)
{
    /* strands of a: [a.i, a.i+len-1] and [a.j-len+1, a.j] */
    int as0 = a->i, as1 = a->i + a->len - 1, at0 = a->j - a->len + 1, at1 = a->j;
    int bs0 = b->i, bs1 = b->i + b->len - 1, bt0 = b->j - b->len + 1, bt1 = b->j;
#define OV(x0, x1, y0, y1) ((x0) <= (y1) && (y0) <= (x1))
    return OV(as0, as1, bs0, bs1) || OV(as0, as1, bt0, bt1) || OV(at0, at1, bs0, bs1) || OV(at0, at1, bt0, bt1);
#undef OV
}

static void choose_stems(stemvec *all, double subopt, stemvec *res)
{
    int n = all->n;
    /* :758 stable sort, descending finalscore */
    orc_stem *v = all->v;
    orc_stem *tmp = (orc_stem *)malloc(sizeof(orc_stem) * (size_t)(n + 1));
    for (int w = 1; w < n; w *= 2) {                      /* bottom-up stable merge sort */
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int a = lo, b = mid, k = lo;
            while (a < mid && b < hi) {
                if (v[b].finscore > v[a].finscore) tmp[k++] = v[b++];
                else tmp[k++] = v[a++];
            }
            while (a < mid) tmp[k++] = v[a++];
            while (b < hi) tmp[k++] = v[b++];
        }
        memcpy(v, tmp, sizeof(orc_stem) * (size_t)n);
    }
    free(tmp);
    res->n = 0;
    if (n == 0) return;                                   /* :763 */
    sv_push(res, v[0]);
    double suboptrange = subopt * v[0].finscore;           /* :769 */
    for (int k = 1; k < n; k++) {                         /* :773-787 */
        if (v[k].finscore < suboptrange) return;
        int all_conf = 1;
        for (int r = 0; r < res->n && all_conf; r++)
            if (!shares_base(&v[k], &res->v[r])) all_conf = 0;
        if (all_conf) sv_push(res, v[k]);
    }
}

/* OptimalStems -- SQRNdbnseq.py:792-833 */
static void optimal_stems(const char *seq, int n, const double *boolmat, const double *scoremat,
                          const int *rbps, int nrbps, const orc_stem *rstems, int nrstems,
                          double subopt, const orc_params *p, double *work, stemvec *res)
{
    /* :801 restbps = rbps - bps(rstems) */
    int *rest = (int *)malloc(sizeof(int) * 2 * (size_t)(nrbps + 1));
    int nrest = 0;
    for (int k = 0; k < nrbps; k++) {
        int v = rbps[2 * k], w = rbps[2 * k + 1], found = 0;
        for (int s = 0; s < nrstems && !found; s++) {
            int d = v - rstems[s].i;
            if (d >= 0 && d < rstems[s].len && rstems[s].j - d == w) found = 1;
        }
        if (!found) { rest[2 * nrest] = v; rest[2 * nrest + 1] = w; nrest++; }
    }
    stemvec all = {0};
    annotate(boolmat, scoremat, n, rest, nrest, rstems, nrstems, p->minlen, p->minbpscore, work, &all);
    score_stems(seq, n, &all, rstems, nrstems, p, p->minfinscore);
    choose_stems(&all, subopt, res);
    free(all.v); free(rest);
}

ORC_API int orc_optimal(const char *seq, int n, const double *boolmat, const double *scoremat,
                        const int *rbps, int nrbps, const int *rstems, int nrstems,
                        double subopt, const orc_params *p,
                        int *out_ijl, double *out_bps, double *out_fin, int cap)
{
    orc_stem *rs = (orc_stem *)calloc((size_t)nrstems + 1, sizeof(orc_stem));
    for (int k = 0; k < nrstems; k++) {
        rs[k].i = rstems[3 * k]; rs[k].j = rstems[3 * k + 1]; rs[k].len = rstems[3 * k + 2];
    }
    double *work = (double *)malloc(sizeof(double) * (size_t)n * n + 8);
    stemvec res = {0};
    optimal_stems(seq, n, boolmat, scoremat, rbps, nrbps, rs, nrstems, subopt, p, work, &res);
    int cnt = res.n;
    for (int k = 0; k < cnt && k < cap; k++) {
        out_ijl[3 * k] = res.v[k].i; out_ijl[3 * k + 1] = res.v[k].j; out_ijl[3 * k + 2] = res.v[k].len;
        out_bps[k] = res.v[k].bpscore; out_fin[k] = res.v[k].finscore;
    }
    free(work); free(rs); free(res.v);
    return cnt;
}

/* --------------------------------------------------------------------------
 * Greedy pool loop of SQRNdbnseq -- SQRNdbnseq.py:1102-1199 (mp=False form).
 * Result: finished structures in the order the reference appends them.
 * Flattened into static buffers read back through orc_greedy_get().
 * -------------------------------------------------------------------------- */
typedef struct { orc_stem *stems; int n; } structure;

static structure *g_fin = NULL;
static int g_nfin = 0, g_capfin = 0;
static long g_calls = 0;   /* R = number of OptimalStems evaluations */

static void fin_push(const orc_stem *stems, int n)
{
    if (g_nfin == g_capfin) {
        g_capfin = g_capfin ? 2 * g_capfin : 64;
        g_fin = (structure *)realloc(g_fin, sizeof(structure) * (size_t)g_capfin);
    }
    g_fin[g_nfin].stems = (orc_stem *)malloc(sizeof(orc_stem) * (size_t)(n + 1));
    memcpy(g_fin[g_nfin].stems, stems, sizeof(orc_stem) * (size_t)n);
    g_fin[g_nfin].n = n;
    g_nfin++;
}

ORC_API int orc_greedy(const char *seq, int n, const double *boolmat, const double *scoremat,
                       const int *rbps, int nrbps, const orc_params *p, int poollim)
{
    for (int k = 0; k < g_nfin; k++) free(g_fin[k].stems);
    g_nfin = 0; g_calls = 0;
    double *work = (double *)malloc(sizeof(double) * (size_t)n * n + 8);
    double cursubopt = p->suboptmin;                      /* :1069 */
    double suboptinc = (p->suboptmax - p->suboptmin) / p->suboptsteps;   /* :1071 */
    structure *cur = (structure *)malloc(sizeof(structure));
    int ncur = 1; cur[0].stems = NULL; cur[0].n = 0;      /* :1105 */
    int cursize = 1;
    while (ncur) {                                        /* :1159 */
        if (ncur > cursize) {                             /* :1162-1165 */
            cursize = ncur;
            if (cursubopt < p->suboptmax) cursubopt += suboptinc;
        }
        structure *next = NULL; int nnext = 0, capnext = 0;
        for (int c = 0; c < ncur; c++) {
            if ((double)cur[c].n == p->maxstemnum) {      /* :1170 */
                fin_push(cur[c].stems, cur[c].n);
                continue;
            }
            stemvec res = {0};
            optimal_stems(seq, n, boolmat, scoremat, rbps, nrbps, cur[c].stems, cur[c].n,
                          cursubopt, p, work, &res);      /* :1182 */
            g_calls++;
            if (res.n) {                                  /* :1190-1193 */
                int stopper = cursize >= poollim ? 1 : res.n;
                for (int k = 0; k < stopper; k++) {
                    if (nnext == capnext) {
                        capnext = capnext ? 2 * capnext : 16;
                        next = (structure *)realloc(next, sizeof(structure) * (size_t)capnext);
                    }
                    next[nnext].n = cur[c].n + 1;
                    next[nnext].stems = (orc_stem *)malloc(sizeof(orc_stem) * (size_t)(cur[c].n + 1));
                    if (cur[c].n) memcpy(next[nnext].stems, cur[c].stems, sizeof(orc_stem) * (size_t)cur[c].n);
                    next[nnext].stems[cur[c].n] = res.v[k];
                    nnext++;
                }
            } else {
                fin_push(cur[c].stems, cur[c].n);         /* :1196 */
            }
            free(res.v);
        }
        for (int c = 0; c < ncur; c++) free(cur[c].stems);
        free(cur);
        cur = next; ncur = nnext;
    }
    free(cur); free(work);
    return g_nfin;
}

ORC_API long orc_greedy_calls(void) { return g_calls; }
ORC_API int orc_greedy_nstems(int k) { return g_fin[k].n; }
ORC_API void orc_greedy_get(int k, int *ijl, double *bps, double *fin)
{
    for (int t = 0; t < g_fin[k].n; t++) {
        ijl[3 * t] = g_fin[k].stems[t].i; ijl[3 * t + 1] = g_fin[k].stems[t].j; ijl[3 * t + 2] = g_fin[k].stems[t].len;
        bps[t] = g_fin[k].stems[t].bpscore; fin[t] = g_fin[k].stems[t].finscore;
    }
}

/* --------------------------------------------------------------------------
 * Nussinov + BackTrack -- SQRNalgos.py:44-93, 6-41 (matrix=None form).
 * scores[(v,w)] = -stem[2] for every cell of every stem (later stems
 * overwrite earlier ones, as the dict comprehension does).
 * -------------------------------------------------------------------------- */
ORC_API int orc_nussinov(const char *seq, int n, const int *stems_ijl, const double *stem_score, int nstems,
                         int *out_pairs, int cap)
{
    const int minloop = 3;
    if (n <= 0) return 0;
    double *S = (double *)calloc((size_t)n * n, sizeof(double));
    unsigned char *has = (unsigned char *)calloc((size_t)n * n, 1);
    for (int s = 0; s < nstems; s++)
        for (int b = 0; b < stems_ijl[3 * s + 2]; b++) {
            int v = stems_ijl[3 * s] + b, w = stems_ijl[3 * s + 1] - b;
            S[(size_t)v * n + w] = -stem_score[s]; has[(size_t)v * n + w] = 1;
        }
    double *D = (double *)calloc((size_t)n * n, sizeof(double));
    int *K = (int *)malloc(sizeof(int) * (size_t)n * n);
    for (size_t t = 0; t < (size_t)n * n; t++) K[t] = -2;   /* -2 = key absent */
#define DD(a, b) (((a) < 0 || (b) < 0) ? D[(size_t)(((a) + n) % n) * n + (((b) + n) % n)] : D[(size_t)(a) * n + (b)])
    for (int h = 1; h < n; h++)
        for (int i = 0; i < n - h; i++) {
            int j = i + h;
            int bestk = -1; double bestscorek = 1e9;       /* 10**9 */
            for (int k = i; k < j - 1; k++)
                if (has[(size_t)k * n + j]) {
                    /* D[i, k-1] with k == i is numpy negative indexing D[i, -1] (:74) */
                    double scorek = DD(i, k - 1) + D[(size_t)(k + 1) * n + (j - 1)] + S[(size_t)k * n + j];
                    if (scorek < bestscorek) { bestk = k; bestscorek = scorek; }
                }
            if (bestscorek <= D[(size_t)i * n + (j - 1)]) {
                K[(size_t)i * n + j] = bestk;
                D[(size_t)i * n + j] = bestscorek;
            } else {
                D[(size_t)i * n + j] = D[(size_t)i * n + (j - 1)];
            }
        }
#undef DD
    /* BackTrack(0, N-1): the queue is a set of cells per level; order inside a
     * level does not matter because the result is sorted. */
    int *q = (int *)malloc(sizeof(int) * 2 * (size_t)(n + 2));
    int *nq = (int *)malloc(sizeof(int) * 2 * (size_t)(n + 2));
    int qn = 1, np = 0; q[0] = 0; q[1] = n - 1;
    unsigned char *inq = (unsigned char *)calloc((size_t)n * n, 1);
    while (qn) {
        int nn = 0;
        for (int t = 0; t < qn; t++) {
            int i = q[2 * t], j = q[2 * t + 1];
#define ANYSEP(a, b, res) do { res = 0; for (int _x = (a); _x < (b); _x++) if (_x >= 0 && _x < n && is_sep(seq[_x])) res = 1; } while (0)
#define PUSHQ(a, b) do { if (!inq[(size_t)(a) * n + (b)]) { inq[(size_t)(a) * n + (b)] = 1; nq[2 * nn] = (a); nq[2 * nn + 1] = (b); nn++; } } while (0)
            if (i < 0 || j < 0 || i >= n || j >= n) continue;
            int kk = K[(size_t)i * n + j];
            if (kk != -2) {
                int k = kk, sepf;
                ANYSEP(i + 1, k - 1, sepf);
                if (((k - 1) - i > minloop) || ((k - 1) - i > 0 && sepf)) PUSHQ(i, k - 1);
                ANYSEP(k + 2, j - 1, sepf);
                if (((j - 1) - (k + 1) > minloop) || ((j - 1) - (k + 1) > 0 && sepf)) PUSHQ(k + 1, j - 1);
                if (np < cap) { out_pairs[2 * np] = k; out_pairs[2 * np + 1] = j; }
                np++;
            } else {
                int sepf;
                ANYSEP(i + 1, j - 1, sepf);
                if (((j - 1) - i > minloop) || ((j - 1) - i > 0 && sepf)) PUSHQ(i, j - 1);
            }
        }
        for (int t = 0; t < nn; t++) inq[(size_t)nq[2 * t] * n + nq[2 * t + 1]] = 0;
        memcpy(q, nq, sizeof(int) * 2 * (size_t)nn);
        qn = nn;
    }
    free(S); free(has); free(D); free(K); free(q); free(nq); free(inq);
    return np;
}
