/* Plain C consumer of include/m2s.h: what a cgo / JNI / Rust-FFI binding links against.  No torch, no Python:
 * host pointers in, host pointers out.  Checks the reference's documented known answers
 * (mesh_to_sdf/src/lib.rs:13-31 generate_sdf doc-test: distance 1.0; lib.rs:37-58 grid doc-test) and a
 * save_to_file / read_from_file round trip.  Exit code 0 = all good; prints one line per check. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "m2s.h"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++fails; printf("FAIL: "); printf(__VA_ARGS__); printf("  [%s]\n", m2s_last_error()); } else { printf("ok: "); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main(int argc, char** argv) {
  printf("m2s version %d, %d HIP device(s)\n", m2s_version(), m2s_device_count());
  if (m2s_device_count() < 1) { printf("no GPU: nothing to run\n"); return 77; }
  {
    /* m2s_warmup: optional, idempotent; the calls below then find runtime, code objects and context ready */
    int w1 = m2s_warmup(-1, 0, 0), w2 = m2s_warmup(0, (size_t)1 << 20, (size_t)1 << 20);
    uint64_t pb[5] = {0, 16, 32, 48, 64}, nb[5] = {0, 0, 0, 0, 0};
    float cost[4] = {1.0f, 3.0f, 3.0f, 1.0f};
    int rb;
    CHECK(w1 == M2S_OK && w2 == M2S_OK, "m2s_warmup twice");
    CHECK(m2s_warmup(1000, 0, 0) == M2S_ERR_BAD_ARG, "m2s_warmup on a device that does not exist -> M2S_ERR_BAD_ARG");
    rb = m2s_balanced_slabs(64, 4, 4, pb, cost, nb);
    CHECK(rb == M2S_OK && nb[0] == 0 && nb[4] == 64 && nb[1] > 16 && nb[2] == 32 && nb[3] < 48, "m2s_balanced_slabs moves the cuts towards the cheap ends (%d %d %d)",
          (int)nb[1], (int)nb[2], (int)nb[3]);
  }

  /* lib.rs:13-31 */
  const float vertices[9] = {0.f, 1.f, 0.f, 1.f, 2.f, 3.f, 1.f, 3.f, 4.f};
  const uint32_t indices[3] = {0, 1, 2};
  const float query[3] = {0.f, 0.f, 0.f};
  float sdf[1] = {0.f};
  size_t n_out = 0;
  int rc = m2s_generate_sdf(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, query, 1, M2S_ACCEL_BVH, M2S_SIGN_RAYCAST, sdf, &n_out, NULL);
  CHECK(rc == M2S_OK && n_out == 1 && sdf[0] == 1.0f, "generate_sdf doc-test: distance %.9g (want 1)", sdf[0]);

  /* u16 indices, default acceleration method (RtreeBvh) */
  const uint16_t idx16[3] = {0, 1, 2};
  rc = m2s_generate_sdf(vertices, 3, idx16, 3, 2, M2S_TRIANGLE_LIST, query, 1, M2S_ACCEL_RTREE_BVH, M2S_SIGN_RAYCAST, sdf, &n_out, NULL);
  CHECK(rc == M2S_OK && sdf[0] == 1.0f, "u16 indices, RtreeBvh: %.9g", sdf[0]);

  /* lib.rs:37-58: grid over [0,10]^3 (well, bbox 0..10), 2x2x2... the doc-test uses cell_count [3,3,3] */
  const float bmin[3] = {0.f, 0.f, 0.f}, bmax[3] = {10.f, 10.f, 10.f};
  const uint64_t count[3] = {3, 3, 3};
  m2s_grid grid;
  m2s_grid_from_bounding_box(bmin, bmax, count, &grid);
  float cells[27];
  rc = m2s_generate_grid_sdf(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, &grid, M2S_SIGN_RAYCAST, cells, NULL);
  CHECK(rc == M2S_OK, "generate_grid_sdf 3x3x3 returns OK");
  /* every cell must equal the generic path on its centre (generate/grid.rs:693-724 asserts exactly this) */
  int same = rc == M2S_OK;
  for (uint64_t x = 0; x < 3 && same; ++x)
    for (uint64_t y = 0; y < 3; ++y)
      for (uint64_t z = 0; z < 3; ++z) {
        const uint64_t c[3] = {x, y, z};
        float p[3], d;
        m2s_grid_cell_center(&grid, c, p);
        rc = m2s_generate_sdf(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, p, 1, M2S_ACCEL_BVH, M2S_SIGN_RAYCAST, &d, NULL, NULL);
        if (rc != M2S_OK || memcmp(&d, &cells[m2s_grid_cell_idx(&grid, c)], 4) != 0) same = 0;
      }
  CHECK(same, "grid cells == generate_sdf on the cell centres (bitwise)");

  /* panics of the reference -> error codes */
  const uint32_t bad[3] = {0, 1, 7};
  rc = m2s_generate_sdf(vertices, 3, bad, 3, 4, M2S_TRIANGLE_LIST, query, 1, M2S_ACCEL_BVH, M2S_SIGN_RAYCAST, sdf, NULL, NULL);
  CHECK(rc == M2S_ERR_BAD_ARG, "out-of-range index -> M2S_ERR_BAD_ARG (%d)", rc);
  rc = m2s_generate_sdf(vertices, 0, NULL, 0, 4, M2S_TRIANGLE_LIST, query, 1, M2S_ACCEL_RTREE, M2S_SIGN_NORMAL, sdf, NULL, NULL);
  CHECK(rc == M2S_ERR_EMPTY_MESH, "Rtree on an empty mesh -> M2S_ERR_EMPTY_MESH (%d)", rc);

  /* persistent mesh */
  m2s_mesh* mesh = NULL;
  rc = m2s_mesh_create(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, NULL, &mesh);
  float cells2[27];
  if (rc == M2S_OK) rc = m2s_mesh_generate_grid_sdf(mesh, &grid, M2S_SIGN_RAYCAST, cells2, NULL);
  CHECK(rc == M2S_OK && memcmp(cells, cells2, sizeof(cells)) == 0 && m2s_mesh_triangle_count(mesh) == 1, "persistent mesh == one-shot");
  m2s_mesh_destroy(mesh);

  /* the same call spread over a device list (here: this GPU three times = three shards, three host threads inside the
   * library); host pointers, every shard writes its x-slab straight into the caller's array */
  {
    const int32_t devs[3] = {0, 0, 0};
    m2s_multi_opts mo;
    memset(&mo, 0, sizeof(mo));
    mo.struct_size = sizeof(mo);
    mo.n_devices = 3;
    mo.devices = devs;
    mo.mem_kind = M2S_MEM_HOST;
    int used = -1;
    mo.exchange_used = &used;
    float cells3[27];
    float* outs[1] = {cells3};
    rc = m2s_generate_grid_sdf_multi(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, &grid, M2S_SIGN_RAYCAST, outs, &mo);
    uint64_t a, b;
    m2s_slab_bounds(3, 3, 1, &a, &b);
    CHECK(rc == M2S_OK && memcmp(cells, cells3, sizeof(cells)) == 0 && used == M2S_XCHG_NONE && a == 1 && b == 2,
          "m2s_generate_grid_sdf_multi over {0,0,0} == single call");
    /* generate_sdf over the same three shards: two queries, so one shard has nothing to do */
    {
      const float qs[6] = {0.25f, 0.25f, 1.0f, 0.25f, 0.25f, -2.0f};
      float one[2] = {0, 0}, multi[2] = {0, 0};
      size_t n1 = 0, n3 = 0;
      float* qouts[1] = {multi};
      rc = m2s_generate_sdf(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, qs, 2, M2S_ACCEL_RTREE_BVH, M2S_SIGN_RAYCAST, one, &n1, NULL);
      CHECK(rc == M2S_OK && n1 == 2, "m2s_generate_sdf");
      rc = m2s_generate_sdf_multi(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, qs, 2, M2S_ACCEL_RTREE_BVH, M2S_SIGN_RAYCAST, qouts, &n3, &mo);
      CHECK(rc == M2S_OK && n3 == 2 && memcmp(one, multi, sizeof(one)) == 0, "m2s_generate_sdf_multi over {0,0,0} == single call");
    }
    /* a version-0.1 caller's options (struct_size = M2S_OPTS_V1_SIZE) still work */
    m2s_opts o1;
    memset(&o1, 0, sizeof(o1));
    o1.struct_size = M2S_OPTS_V1_SIZE;
    o1.device = -1;
    o1.synchronous = 1;
    o1.lane = 12345;   /* garbage beyond the v1 size must be ignored */
    rc = m2s_generate_grid_sdf(vertices, 3, indices, 3, 4, M2S_TRIANGLE_LIST, &grid, M2S_SIGN_RAYCAST, cells3, &o1);
    CHECK(rc == M2S_OK && memcmp(cells, cells3, sizeof(cells)) == 0, "m2s_opts of version 0.1 accepted");
  }

  /* container round trip through a file */
  const char* path = argc > 1 ? argv[1] : "/tmp/m2s_abi_smoke.bin";
  rc = m2s_sdf_save_grid(path, &grid, cells, 27, NULL);
  m2s_sdf_info info;
  if (rc == M2S_OK) rc = m2s_sdf_probe_file(path, &info);
  float back[27];
  if (rc == M2S_OK) rc = m2s_sdf_read_file(path, NULL, back, NULL);
  CHECK(rc == M2S_OK && info.kind == M2S_SDF_GRID && info.n_distances == 27 && memcmp(&info.grid, &grid, sizeof(grid)) == 0 &&
            memcmp(cells, back, sizeof(cells)) == 0, "save_to_file / read_from_file round trip (%zu bytes)", m2s_sdf_grid_encoded_size(&grid, 27));
  remove(path);

  /* cell ordering */
  uint32_t order[27];
  float lim[2];
  rc = m2s_order_cells_by_distance(cells, 27, order, lim, NULL);
  int sorted = rc == M2S_OK;
  for (int i = 1; i < 27 && sorted; ++i) sorted = cells[order[i - 1]] <= cells[order[i]];
  CHECK(sorted && lim[0] == cells[order[0]] && lim[1] == cells[order[26]], "cells ordered by distance, iso limits [%g, %g]", lim[0], lim[1]);

  m2s_release_workspace();
  printf(fails ? "%d check(s) FAILED\n" : "all checks passed\n", fails);
  return fails ? 1 : 0;
}
