/* oracle_cli - command line front end of the CPU oracle (TEST INFRASTRUCTURE).
 *   oracle_cli sizes <file.dat>
 *   oracle_cli solve <file.dat> [gap] [verbose]
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s sizes|solve file.dat [gap] [verbose]\n", argv[0]); return 2; }
  char err[256];
  oinst* I = orc_from_dat(argv[2], err, sizeof(err));
  if (!I) { fprintf(stderr, "%s\n", err); return 1; }
  if (!strcmp(argv[1], "sizes")) {
    orc_sizes s = orc_raw_sizes(I);
    printf("rows %d bin %d cont %d nnz %d\n", s.rows, s.bin, s.cont, s.nnz);
  } else {
    orc_opts o = {argc > 3 ? atof(argv[3]) : -1.0, 0.0, 0, argc > 4 ? atoi(argv[4]) : 0};
    miqp_raw_results_c* r = orc_results_alloc(I->C, I->N, I->R, I->E, I->O, I->L);
    miqp_solution_properties_c p;
    int st = orc_solve(I, &o, r, &p);
    printf("status %d cpx %d objective %.10g bound %.10g gap %.3g nodes %lld iters %d time %.3f\n", st, p.status,
           p.objective, p.best_bound, p.gap, p.nodes, p.NrIterations, p.time);
    if (st == 0) {
      double obj; char w[96];
      double v = orc_raw_eval(I, r, r->slackvars_real, &obj, w, sizeof(w));
      printf("raw-model check: max violation %.3e (%s) objective %.10g\n", v, w, obj);
    }
    orc_results_free(r);
  }
  orc_free(I);
  return 0;
}
