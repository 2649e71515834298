// sq_context.h -- ScoreStems' sweep over the strands between a candidate's innermost pair (SQRNdbnseq.py:665-689) in
// closed form for structures without crossing stems.
//
// The reference walks the positions sa+1 .. sb-1 and books, in order: sub-ECR faces ("blocks": a paired region whose two
// ends lie inside the span; their unpaired positions do not count as dots, the first one defines the internal loop),
// and wing strands (their partner lies outside the span: the candidate would cross them).  The scoring kernel's strand walk
// does the same over the sorted strand list with skip pointers over the blocks, which on long sequences still is a
// chain of dozens of dependent reads per candidate -- the closing strands of every helix around sa, the sibling
// helices of every loop on the way, the opening strands of every helix around sb.
//
// For a structure whose stems do not cross (every strand on level 1) the strands form a forest, and that walk is a
// path in it: from sa rightwards up to the loop that holds both ends (crossing closers = wings, skipping sibling
// subtrees = blocks), then down to sb -- which read from sb LEFTWARDS is the same kind of climb (crossing openers =
// wings, skipping subtrees = blocks).  With g = 0 .. S the gaps between the S sorted strands (gap g lies before strand g):
//     depth[g]            strands opened and not closed before gap g
//     right chain         next(g) = gap behind the partner of strand g (an opener: a block) or g + 1 (a closer: a wing)
//     left chain          prev(g) = gap before the partner of strand g - 1 (a closer: a block) or g - 1 (an opener: a wing)
//     R*[g], L*[g]        blocks / unpaired positions inside them / wing lengths summed along the chain from g to the end
// the two climbs of a candidate whose span covers the gaps a .. b meet at g* = the rightmost gap of minimum depth in
// [a, b] (a range-minimum query on a sparse table), and
//     blocks   = (RB[a] - RB[g*]) + (LB[b] - LB[g*])      (likewise the covered positions and the wing lengths)
//     first block (only used when there is exactly one): the first opener at or behind a, or the partner of the last
//     closer before b.
// sq_context_kernel builds the tables once per structure and round (pointer jumping: log S steps); the scoring kernel
// answers every candidate with two binary searches and seven table reads.  Integer arithmetic only: the results are the
// walk's, bit for bit (SQ_CTX_CHECK builds run both and compare).  Structures with crossing stems keep the walk.
#pragma once
#include <hip/hip_runtime.h>
#include "sq_internal.h"

struct SqCtxRec { uint16_t rb, rc, rw, lb, lc, lw, fr, fl; };   // one gap: chain sums to the right / left, first blocks

struct SqCtxTab {
    SqCtxRec *rec;       // [state slot of the structure][cap]
    int16_t *depth;      // [slot][cap]
    uint16_t *rmq;       // [slot][levels][cap]: rightmost argmin of depth over [g, g + 2^j - 1], j = 1 .. levels
    uint8_t *ok;         // [slot] 1: the tables are valid (no crossing stems, the strands fit)
    int32_t cap, levels; // gaps per structure (strands + 1), sparse-table levels; rec == nullptr: no tables in this launch
};

struct SqCtxOut { int nrec, be0, be1, covered, brackets; };

// the sweep of a candidate whose span holds the strands a .. b - 1 (a < b), S: the structure's sorted strands
__device__ __forceinline__ void sq_ctx_query(const SqCtxRec *rec, const int16_t *depth, const uint16_t *rmq, int cap,
                                             const SqStrand *S, int a, int b, SqCtxOut &o)
{
    const int j = 31 - __clz(b - a + 1);                             // >= 1
    const int c1 = rmq[(size_t)(j - 1) * cap + a], c2 = rmq[(size_t)(j - 1) * cap + b - (1 << j) + 1];
    const int d1 = depth[c1], d2 = depth[c2];
    const int g = d2 < d1 ? c2 : (d1 < d2 ? c1 : (c1 > c2 ? c1 : c2));
    const SqCtxRec ra = rec[a], rg = rec[g], rb = rec[b];
    const int nbr = (int)ra.rb - (int)rg.rb, nbl = (int)rb.lb - (int)rg.lb;
    o.nrec = nbr + nbl;
    o.covered = ((int)ra.rc - (int)rg.rc) + ((int)rb.lc - (int)rg.lc);
    o.brackets = ((int)ra.rw - (int)rg.rw) + ((int)rb.lw - (int)rg.lw);
    o.be0 = 0; o.be1 = 0;
    if (o.nrec == 1) {
        const SqStrand x = S[nbr ? ra.fr : rb.fl];
        o.be0 = x.start; o.be1 = x.pstart;
    }
}

size_t sq_context_bytes_per_gap(int cap, int *levels);
void sq_launch_context(const SqStruct *structs, const SqStrand *strands, const SqCtxTab &t, int S, hipStream_t st);
