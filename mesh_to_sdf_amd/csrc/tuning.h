// tuning.h — every run-time knob of the library, in one table.
//
// The values are read from the environment ONCE, when the library is first used (tuning()), by the single getenv loop in
// tuning.cpp; a process that wants to change one afterwards — the tests that force a walk flavour, tools/soak_modes.py — calls
// m2s_tuning_set(name, value) (include/m2s.h).  None of them changes a result: they select between code paths that produce the
// same bits (that is what the tests use them for) or move a crossover.  DESIGN.md §9 carries the same table with the
// measurements behind the defaults.
#pragma once
#include <stdint.h>

namespace m2s {

struct Tuning {
  // ---- diagnostics
  int stats = 0;                    // M2S_STATS        1: traversal counters of the packet walk on stderr, 2: + a counting pass from the final bound
  int host_times = 0;               // M2S_HOST_TIMES   1: where the host time of a call goes, on stderr
  // ---- which walk (tests force each flavour; the defaults are measured crossovers, distance.hip)
  int lane_walk = -1;               // M2S_LANE_WALK    -1 automatic, 0 never, 1 always: one voxel / query per lane instead of one packet per wave
  double lane_ratio = 60.0;         // M2S_LANE_RATIO   grid: lane walk above this many triangles per packet brick
  double lane_ratio_split = 100.0;  // M2S_LANE_RATIO_SPLIT   ... and above this many where the packet walk's stragglers can be split (below)
  double query_lane_coeff = 2.5;    // M2S_QUERY_LANE_COEFF   queries: lane walk below this many queries per triangle
  int treelets = -1;                // M2S_TREELETS     the build's treelet pass (groups of <= 64 triangles re-partitioned by sweep splits): 1 always, 0 never, -1 unless the leaves hold 8 or more triangles (4 or more below 32 768 triangles) or a one-shot grid call walks fewer than 3 M cells (32 M below 4 096 triangles)
  int sort_tile = -1;               // M2S_SORT_TILE    the build's sort of the (key, triangle) pairs: -1 automatic (sample sort with tiles of 1024 / 2048 pairs up to 122 880 / 229 376 triangles, rocPRIM above), 0 rocPRIM always, 1024 / 2048 that tile size where it can hold the mesh
  uint32_t leaf_max = 0;            // M2S_LEAF_MAX     triangles per collapsed leaf of the tree a grid call walks (1 .. 16; a resident tree is re-marked); 0: by triangles per brick (2 / 4 / 8 / 16)
  double brute_max = -1.0;          // M2S_BRUTE_MAX    tree-less path for cells x triangles (queries x triangles) up to this; < 0: automatic, 0: never
  uint32_t cut_min_packets = 100000;// M2S_CUT_MIN_PACKETS   grid: cut lists from this many packets on
  uint32_t query_cut_min = 20000;   // M2S_QUERY_CUT_MIN     queries: cut lists from this many packets on
  int query_launch_tight = 0;       // M2S_QUERY_LAUNCH_TIGHT  test hook: forces the consecutive-packet fallback of the query packets
  int group = -1;                   // M2S_GROUP        packets as workgroups of 2 or 4 waves (launches shallower than the chip): -1 automatic, 0 never, 1 always (where the walk form allows it; four waves)
  uint32_t group_target_waves = 32768; // M2S_GROUP_TARGET_WAVES  ... as many waves per packet (2 or 4) as keep the launch within this many waves
  double group_min_ratio = 1.0;     // M2S_GROUP_MIN_RATIO     ... and from this many triangles per packet brick on
  // ---- cut lists
  float cut_near = 2.0f;            // M2S_CUT_NEAR     emission radius of a list entry, in brick radii (next to the surface)
  float cut_far = 1.0f / 32.0f;     // M2S_CUT_FAR      ... and as a fraction of the distance (far from it)
  int cut_coarse = -1;              // M2S_CUT_COARSE   grid cut lists in two levels (a coarse pass per 4 x 4 x 4 bricks first): -1 automatic (from M2S_CUT_COARSE_MIN_WAVES fine waves on), 0 never, 1 always
  uint32_t cut_coarse_min_waves = 40000;   // M2S_CUT_COARSE_MIN_WAVES   fine k_cut waves (blocks of 4 x 4 x 4 bricks) from which the automatic choice takes two levels (512^3: 32 768 waves, a wash; 1024^3: 262 144, - 1.5 %)
  uint32_t cut_coarse_cap = 0;      // M2S_CUT_COARSE_CAP   node visits after which a coarse-level wave emits what it meets; 0: as M2S_CUT_WAVE_CAP
  uint32_t cut_wave_cap = 0;        // M2S_CUT_WAVE_CAP node visits after which a k_cut wave emits what it meets; 0: max(120, 20 x tree depth)
  // ---- heavy packets (distance.hip "split walk")
  int split = -1;                   // M2S_SPLIT        -1 automatic, 0 never, 1 on, 2 on with the "out of work" flags raised from the start (tests): walks still running when the launch runs dry hand subtrees to other waves
  uint32_t split_budget = 0;        // M2S_SPLIT_BUDGET work units (3 per leaf + 1 per pre-test + 4 per exact evaluation) a walk does before it first looks whether its XCD is running dry; 0: 128
  double split_patience = 1.5;      // M2S_SPLIT_PATIENCE   a packet still walking this many ordinary packet times (measured by the launch) after its XCD ran out of packets is suspended
  uint32_t split_min_records = 16;  // M2S_SPLIT_MIN_RECORDS  a suspended walk hands over surviving subtrees of at least this many node records ...
  uint32_t split_max_records = 128; // M2S_SPLIT_MAX_RECORDS  ... and at most this many (larger ones it opens itself)
  uint32_t split_rounds = 2;        // M2S_SPLIT_ROUNDS follow-up launches: the continuations of the suspended packets, then their subtrees (with more rounds, subtrees may be suspended in their turn; the last round walks to the end)
  int split_report = 0;             // M2S_SPLIT_REPORT 1: suspended packets and items per round of every grid walk, on stderr (synchronises)
  int defer = -1;                   // M2S_DEFER        the packet walk's leaf work: -1 automatic (by triangles per brick: 2 / 1 / 3), 0 wave-wide at once, 1 exact evaluations queued as (voxel, triangle) pairs and run 64 at a time, 2 + wave-wide where >= 48 lanes are reached, 3 the pre-tests queued too
  // ---- host-pointer calls and peer delivery
  uint32_t host_piece_mb = 32;      // M2S_HOST_PIECE_MB   x-pieces of the result streamed to the host while the next is walked
  uint32_t push_pieces = 4;         // M2S_PUSH_PIECES     x-pieces of a slab pushed to the peers while the next is walked
  uint32_t push_blocks = 0;         // M2S_PUSH_BLOCKS     workgroups of a push kernel; 0: 256 (pieces) / 64 (trailing)
};

const Tuning& tuning();                                   // loaded from the environment on first use
int tuning_set(const char* name, const char* value);      // value NULL: back to the default; 0 ok, -1 unknown name / unparsable value
int tuning_describe(char* buf, int cap);                  // "NAME=value\n" lines of the current values; returns the length needed

}  // namespace m2s
