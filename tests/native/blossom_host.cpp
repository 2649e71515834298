// Test-only host harness: runs the product's blossom restatement (squarna_amd/csrc/sq_blossom.h)
// on the CPU so tests can compare it with networkx.max_weight_matching on many graphs quickly.
// stdin: T, then per graph: n m, then m lines "v w weight" (vertex ids in graph order).
// stdout: per graph one line with mate[0..n-1], then the rank of every vertex's first mate assignment (mord[0..n-1]).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../squarna_amd/csrc/sq_blossom.h"

int main()
{
    int T;
    if (scanf("%d", &T) != 1) return 1;
    while (T--) {
        int n, m;
        if (scanf("%d %d", &n, &m) != 2) return 1;
        std::vector<SqMatchEdge> e(m);
        for (int k = 0; k < m; k++) {
            if (scanf("%d %d %lf", &e[k].v, &e[k].w, &e[k].weight) != 3) return 1;
        }
        std::vector<char> scratch(SqBlossom::scratch_bytes(n, m) + 64);
        SqBlossom bl;
        bl.init(n, m, e.data(), scratch.data());
        bl.run();
        if (bl.error) { printf("ERROR %d\n", bl.error); continue; }
        for (int v = 0; v < n; v++) printf("%d ", bl.mate[v]);
        for (int v = 0; v < n; v++) printf("%d ", bl.mord[v]);
        printf("\n");
    }
    return 0;
}
