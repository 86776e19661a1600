/* A caller of include/aslr_to_amd.h written in plain C against the HIP runtime: no Python, no torch in the process.
 * TEST INFRASTRUCTURE (tests/test_gpu_c_abi.py builds it with gcc on the GPU box and compares its output with the
 * Python layer's, bit for bit).
 *
 *   solve_from_c PROBLEM.bin OUT.bin
 *
 * PROBLEM.bin (written by the test from a lowered problem):
 *   aslr_problem_desc_t (pointers ignored) | aslr_solver_params_t | int32 node_model[T+1] | double x0[B][nx] |
 *   int32 has_frame_ref | double frame_ref[B][12] (if has_frame_ref)
 * OUT.bin: double xs[T+1][B][nx] | double us[T][B][nu] | double cost[B] | int32 iter[B] | int32 status[B] | int32 batch_iters
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aslr_to_amd.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "solve_from_c: %s failed (line %d): %s\n", #c, __LINE__, aslr_last_error()); return 1; } } while (0)

static int read_all(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: solve_from_c PROBLEM.bin OUT.bin\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  CHECK(f != NULL);
  aslr_problem_desc_t *desc = (aslr_problem_desc_t *)malloc(sizeof *desc);
  aslr_solver_params_t sp;
  CHECK(read_all(f, desc, sizeof *desc) && read_all(f, &sp, sizeof sp));
  const int B = desc->B, T = desc->T, nx = 4 * desc->chain.nj, nu = desc->models[0].nu;
  int32_t *node_model = (int32_t *)malloc(sizeof(int32_t) * (T + 1));
  double *x0 = (double *)malloc(sizeof(double) * B * nx), *frame_ref = NULL;
  int32_t has_ref = 0;
  CHECK(read_all(f, node_model, sizeof(int32_t) * (T + 1)) && read_all(f, x0, sizeof(double) * B * nx) && read_all(f, &has_ref, 4));
  if (has_ref) {
    frame_ref = (double *)malloc(sizeof(double) * B * 12);
    CHECK(read_all(f, frame_ref, sizeof(double) * B * 12));
  }
  fclose(f);
  desc->node_model = node_model; desc->x0 = x0; desc->frame_ref = frame_ref;

  CHECK(aslr_abi_version() == ASLR_ABI_VERSION);
  const int64_t bytes = aslr_workspace_bytes(desc);
  CHECK(bytes > 0);
  void *ws = NULL;                          /* the caller owns the one device buffer */
  hipStream_t stream;
  CHECK(hipMalloc(&ws, (size_t)bytes) == hipSuccess);
  CHECK(hipStreamCreate(&stream) == hipSuccess);
  aslr_problem_t *p = NULL;
  CHECK(aslr_problem_create(desc, ws, bytes, stream, &p) == ASLR_OK);

  /* cold start: solver.solve([], [], maxiter) -> xs = 0, us = 0 */
  aslr_region_t rx, ru, rf, ri;
  CHECK(aslr_problem_region(p, ASLR_R_XS, &rx) == ASLR_OK && aslr_problem_region(p, ASLR_R_US, &ru) == ASLR_OK);
  CHECK(aslr_problem_region(p, ASLR_R_TRAJ_F, &rf) == ASLR_OK && aslr_problem_region(p, ASLR_R_TRAJ_I, &ri) == ASLR_OK);
  CHECK(hipMemsetAsync((char *)ws + rx.offset, 0, (size_t)rx.bytes, stream) == hipSuccess);
  CHECK(hipMemsetAsync((char *)ws + ru.offset, 0, (size_t)ru.bytes, stream) == hipSuccess);
  int32_t batch_iters = 0;
  CHECK(aslr_solve(p, &sp, 4, stream, &batch_iters) == ASLR_OK);
  CHECK(hipStreamSynchronize(stream) == hipSuccess);

  double *xs = (double *)malloc((size_t)rx.bytes), *us = (double *)malloc((size_t)ru.bytes);
  double *tf = (double *)malloc((size_t)rf.bytes);
  int32_t *ti = (int32_t *)malloc((size_t)ri.bytes);
  CHECK(hipMemcpy(xs, (char *)ws + rx.offset, (size_t)rx.bytes, hipMemcpyDeviceToHost) == hipSuccess);
  CHECK(hipMemcpy(us, (char *)ws + ru.offset, (size_t)ru.bytes, hipMemcpyDeviceToHost) == hipSuccess);
  CHECK(hipMemcpy(tf, (char *)ws + rf.offset, (size_t)rf.bytes, hipMemcpyDeviceToHost) == hipSuccess);
  CHECK(hipMemcpy(ti, (char *)ws + ri.offset, (size_t)ri.bytes, hipMemcpyDeviceToHost) == hipSuccess);
  FILE *o = fopen(argv[2], "wb");
  CHECK(o != NULL);
  fwrite(xs, 1, sizeof(double) * (size_t)(T + 1) * B * nx, o);
  fwrite(us, 1, sizeof(double) * (size_t)T * B * nu, o);
  fwrite(tf + (size_t)ASLR_TF_COST * B, sizeof(double), B, o);
  fwrite(ti + (size_t)ASLR_TI_ITER * B, sizeof(int32_t), B, o);
  fwrite(ti + (size_t)ASLR_TI_STATUS * B, sizeof(int32_t), B, o);
  fwrite(&batch_iters, sizeof batch_iters, 1, o);
  fclose(o);
  CHECK(aslr_problem_destroy(p) == ASLR_OK);
  hipFree(ws);
  hipStreamDestroy(stream);
  printf("solve_from_c: B=%d T=%d nx=%d nu=%d, %d lock-step iterations, cost[0] %.12e\n", B, T, nx, nu, batch_iters,
         tf[(size_t)ASLR_TF_COST * B]);
  return 0;
}
