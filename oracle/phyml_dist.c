/* phyml_dist.c -- TEST INFRASTRUCTURE, development container only: prints the pairwise ML distance
 * matrix the reference's bundled PhyML feeds to BIONJ (the call sequence of src/treeML.c:52-112 up
 * to ML_Dist / Fill_Missing_Dist), for one block in the harness input format.  Used to debug
 * rnacode_amd/csrc/rc_tree.cpp against PhyML. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "utilities.h"
#include "lk.h"
#include "models.h"
#include "free.h"
#include "options.h"
#include "bionj.h"

seq **Get_Seq_local(seq **, option *io, int rw);

int main(int argc, char **argv) {
  FILE *fp = fopen(argv[1], "r");
  int want = argc > 2 ? atoi(argv[2]) : 0, cur = -1, n = 0, i, j;
  char *line = NULL; size_t cap = 0;
  char *names[600], *seqs[600];
  while (getline(&line, &cap, fp) > 0) {
    if (line[0] == 'B') { cur++; n = 0; }
    else if (line[0] == 'S' && cur == want) {
      char *nm = malloc(strlen(line)), *sq = malloc(strlen(line)); int a, b, d; char st;
      sscanf(line + 2, "%s %d %d %c %d %s", nm, &a, &b, &st, &d, sq);
      for (i = 0; sq[i]; i++) sq[i] = toupper(sq[i]);
      names[n] = nm; seqs[n] = sq; n++;
    } else if (line[0] == 'E' && cur == want) break;
  }
  {
    option *io = (option *)Make_Input();
    seq **data; allseq *alldata; matrix *mat; model *mod; int L = strlen(seqs[0]);
    Set_Defaults_Input(io); Set_Defaults_Model(io->mod); Set_Defaults_Optimiz(io->mod->s_opt);
    Make_Model_Complete(io->mod); mod = io->mod;
    io->mod->datatype = NT; io->mod->s_opt->print = 0; io->mod->s_opt->opt_topo = 0; io->mod->s_opt->opt_bl = 1;
    io->mod->s_opt->opt_num_param = 1; io->mod->s_opt->opt_kappa = 1; io->mod->kappa = 4.0; io->mod->n_otu = n;
    data = (seq **)mCalloc(n, sizeof(seq *));
    for (i = 0; i < n; i++) {
      data[i] = (seq *)mCalloc(1, sizeof(seq)); data[i]->len = L;
      data[i]->name = (char *)mCalloc(T_MAX_NAME, sizeof(char)); strcpy(data[i]->name, names[i]);
      data[i]->state = (char *)mCalloc(T_MAX_SEQ, sizeof(char)); strcpy(data[i]->state, seqs[i]);
      data[i]->is_ambigu = NULL;
    }
    data = Get_Seq_local(data, io, 0);
    alldata = Compact_Seq(data, io);
    Free_Seq(data, alldata->n_otu);
    Check_Ambiguities(alldata, io->mod->datatype, io->mod->stepsize);
    Init_Model(alldata, mod);
    printf("freqs %.10f %.10f %.10f %.10f\n", mod->pi[0], mod->pi[1], mod->pi[2], mod->pi[3]);
    mat = ML_Dist(alldata, mod);
    Fill_Missing_Dist(mat);
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) printf("%.8f ", mat->dist[i][j]); printf("\n"); }
  }
  return 0;
}
