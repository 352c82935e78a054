/* oracle_sanitize.c -- drives every oracle entry point once; built with -fsanitize=address,undefined by
 * tests/test_oracle_sanitize.py (sanitizers run on the CPU build only: the GPU pool has no ASan). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/ip_oracle.h"

int main(void)
{
    double v3[ORC3_M], v4[ORC4_M], d[11], pos[66], acc[4];
    orc_step_info info;
    int i;
    orc_init_default(3, v3);
    for (i = 0; i < 50; ++i) orc_step_dir(3, v3, d, &info);
    orc_init_stuck_f3(v3);
    for (i = 0; i < 5; ++i) orc_step(3, v3, &info);
    orc_init_default(4, v4);
    for (i = 0; i < 50; ++i) orc_step(4, v4, NULL);
    orc_sample_trajectory(3, v3, pos, acc);
    orc_sample_trajectory(4, v4, pos, acc);
    orc_init_default(3, v3);
    v3[ORC3_DUR0] = 1.0; v3[ORC3_DUR1] = 1.2; v3[ORC3_VEL1] = 150.0;      /* several violated constraints */
    orc_move_toward_feasibility(3, v3);
    orc_init_default(4, v4);
    v4[ORC4_DUR0] = 1.0;
    orc_move_toward_feasibility(4, v4);
    {
        const size_t n = 2000;
        double *p0 = malloc(n * 8), *p1 = malloc(n * 8), *p2 = malloc(n * 8), *aos = malloc(n * ORC3_M * 8);
        int32_t *it = malloc(n * 4);
        int dist;
        for (dist = 0; dist < 3; ++dist) {
            orc_gen_problems(7, 5, n, dist, p0, p1, p2);
            orc_batch_init_feasible(3, n, p0, p1, p2, aos);
            orc_batch_steps(3, n, aos, 2, 2);
            printf("dist %d total %lld\n", dist, (long long)orc_batch_solve_gated(3, n, aos, 1e-8, 200, it, 3));
        }
        orc_batch_init_feasible(4, n, p0, p1, p2, aos);
        orc_batch_steps(4, n, aos, 3, 0);
        free(p0); free(p1); free(p2); free(aos); free(it);
    }
    {   /* degenerate inputs must not read or write out of bounds either */
        double A[16 * 16] = {0}, b[16] = {0}, x[16];
        orc_colpiv_qr_solve(1, A, b, x);
        orc_colpiv_qr_solve(16, A, b, x);
        memset(v3, 0, sizeof v3);
        orc_step(3, v3, &info);              /* all-zero state: divisions by zero, NaNs, 100 halvings */
    }
    puts("ok");
    return 0;
}
