// The reference crate's own test assertions, written against the C++ mirror (include/mesh_to_sdf.hpp).
// Each block cites the test or doc-test it restates.  Needs a GPU (every compute call runs the HIP kernels).
//   argv[1] = directory with the golden files (tests/golden), argv[2] = scratch directory.
#include <array>
#include <cstdio>
#include <string>
#include <vector>

#include "mesh_to_sdf.hpp"

using namespace mesh_to_sdf;
using P = std::array<float, 3>;

struct Vec3 {   // a cgmath / glam-like point type: x, y, z members
  float x, y, z;
  Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  bool operator==(const Vec3& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct Pt {     // a type with accessor functions and padding (not 12 bytes): must be packed
  double pad;
  float a, b, c;
  Pt(float x_, float y_, float z_) : pad(0), a(x_), b(y_), c(z_) {}
  float x() const { return a; }
  float y() const { return b; }
  float z() const { return c; }
};

static int fails = 0, checks = 0;
#define EXPECT(cond) do { ++checks; if (!(cond)) { ++fails; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

static std::vector<uint8_t> slurp(const std::string& path) {
  std::vector<uint8_t> v;
  if (FILE* f = std::fopen(path.c_str(), "rb")) {
    int c;
    while ((c = std::fgetc(f)) != EOF) v.push_back((uint8_t)c);
    std::fclose(f);
  }
  return v;
}

int main(int argc, char** argv) {
  const std::string gold = argc > 1 ? argv[1] : "tests/golden", tmp = argc > 2 ? argv[2] : "/tmp";

  {   // crate doc-test, lib.rs:10-58
    std::vector<P> vertices = {{0.5f, 1.5f, 0.5f}, {1.f, 2.f, 3.f}, {1.f, 3.f, 7.f}};
    std::vector<uint32_t> indices = {0, 1, 2};
    std::vector<P> query_points = {{0.5f, 0.5f, 0.5f}};
    auto sdf = generate_sdf(vertices, Topology<uint32_t>::TriangleList(indices), query_points, AccelerationMethod::RtreeBvh());
    EXPECT(sdf == std::vector<float>{1.0f});
    auto grid = Grid<P>::from_bounding_box({0.f, 0.f, 0.f}, {10.f, 10.f, 10.f}, {10, 10, 10});
    auto gsdf = generate_grid_sdf(vertices, Topology<uint32_t>::TriangleList(indices), grid, SignMethod::Raycast);
    EXPECT(gsdf.size() == 1000 && gsdf[0] == 1.0f);
  }
  {   // generate_sdf doc-test, lib.rs:265-289, through every point / index flavour
    std::vector<P> v = {{0.f, 1.f, 0.f}, {1.f, 2.f, 3.f}, {1.f, 3.f, 4.f}};
    std::vector<uint32_t> i32 = {0, 1, 2};
    std::vector<uint16_t> i16 = {0, 1, 2};
    std::vector<uint64_t> i64 = {0, 1, 2};
    std::vector<P> q = {{0.f, 0.f, 0.f}};
    EXPECT(generate_sdf(v, Topology<uint32_t>::TriangleList(i32), q, AccelerationMethod::RtreeBvh()) == std::vector<float>{1.0f});
    EXPECT(generate_sdf(v, Topology<uint16_t>::TriangleList(i16), q) == std::vector<float>{1.0f});          // default method
    EXPECT(generate_sdf(v, Topology<uint64_t>::TriangleList(i64), q, AccelerationMethod::Bvh()) == std::vector<float>{1.0f});
    EXPECT(generate_sdf(v, Topology<uint32_t>::TriangleList(), q, AccelerationMethod::None()) == std::vector<float>{1.0f});   // None => 0..len
    EXPECT(generate_sdf(v, Topology<uint32_t>::TriangleStrip(i32), q, AccelerationMethod::Rtree()) == std::vector<float>{1.0f});
    std::vector<Vec3> vv = {{0.f, 1.f, 0.f}, {1.f, 2.f, 3.f}, {1.f, 3.f, 4.f}}, qv = {{0.f, 0.f, 0.f}};
    EXPECT(generate_sdf(vv, Topology<uint32_t>::TriangleList(i32), qv) == std::vector<float>{1.0f});
    std::vector<Pt> vp = {{0.f, 1.f, 0.f}, {1.f, 2.f, 3.f}, {1.f, 3.f, 4.f}}, qp = {{0.f, 0.f, 0.f}};
    EXPECT(generate_sdf(vp, Topology<uint32_t>::TriangleList(i32), qp) == std::vector<float>{1.0f});
  }
  {   // generate/grid.rs:693-724 test_generate_grid: the grid equals generate_sdf on the 125 cell centres
    std::vector<P> vertices = {{0.f, 1.f, 0.f}, {1.f, 2.f, 3.f}, {1.f, 3.f, 4.f}, {2.f, 0.f, 0.f}};
    std::vector<uint32_t> indices = {0, 1, 2, 1, 2, 3};
    auto grid = Grid<P>::from_bounding_box({0.f, 0.f, 0.f}, {5.f, 5.f, 5.f}, {5, 5, 5});
    std::vector<P> query_points;
    for (size_t x = 0; x < grid.get_cell_count()[0]; ++x)
      for (size_t y = 0; y < grid.get_cell_count()[1]; ++y)
        for (size_t z = 0; z < grid.get_cell_count()[2]; ++z) query_points.push_back(grid.get_cell_center({x, y, z}));
    auto sdf = generate_sdf(vertices, Topology<uint32_t>::TriangleList(indices), query_points, AccelerationMethod::None(SignMethod::Raycast));
    auto grid_sdf = generate_grid_sdf(vertices, Topology<uint32_t>::TriangleList(indices), grid, SignMethod::Raycast);
    EXPECT(sdf.size() == 125 && sdf == grid_sdf);
  }
  {   // grid.rs:176-297 unit tests
    auto g = Grid<P>::new_({0.1f, 0.2f, 0.3f}, {1.1f, 1.2f, 1.3f}, {11, 12, 13});
    EXPECT((g.get_first_cell() == P{0.1f, 0.2f, 0.3f}) && (g.get_cell_size() == P{1.1f, 1.2f, 1.3f}) && (g.get_cell_count() == std::array<size_t, 3>{11, 12, 13}));
    g = Grid<P>::new_({0.f, 1.f, 2.f}, {1.f, 2.f, 3.f}, {10, 20, 30});
    EXPECT((g.get_first_cell() == P{0.f, 1.f, 2.f}) && (g.get_last_cell() == P{10.f, 41.f, 92.f}));
    g = Grid<P>::from_bounding_box({-1.f, 0.f, 1.f}, {0.f, 2.f, 5.f}, {2, 2, 2});
    EXPECT((g.get_first_cell() == P{-0.75f, 0.5f, 2.f}) && (g.get_cell_size() == P{0.5f, 1.f, 2.f}));
    EXPECT((g.get_bounding_box() == std::pair<P, P>{{-1.f, 0.f, 1.f}, {0.f, 2.f, 5.f}}));
    g = Grid<P>::from_bounding_box({0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, {2, 2, 2});
    EXPECT((g.snap_point_to_grid({0.4f, 0.8f, 0.1f}) == SnapResult{SnapKind::Inside, {0, 1, 0}}));
    EXPECT((g.snap_point_to_grid({-0.5f, 0.8f, 0.8f}) == SnapResult{SnapKind::Outside, {0, 1, 1}}));
    EXPECT((g.snap_point_to_grid({0.8f, 0.8f, 0.8f}) == SnapResult{SnapKind::Inside, {1, 1, 1}}));
    EXPECT((g.snap_point_to_grid({0.8f, 1.5f, 0.8f}) == SnapResult{SnapKind::Outside, {1, 1, 1}}));
    EXPECT((g.get_cell_center({0, 0, 0}) == P{0.25f, 0.25f, 0.25f}) && (g.get_cell_center({1, 0, 1}) == P{0.75f, 0.25f, 0.75f}));
    g = Grid<P>::from_bounding_box({0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, {2, 3, 4});
    EXPECT(g.get_cell_idx({0, 0, 1}) == 1 && g.get_cell_idx({0, 1, 0}) == 4 && g.get_cell_idx({1, 0, 0}) == 12 && g.get_cell_idx({1, 1, 1}) == 17);
    g = Grid<P>::from_bounding_box({0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, {5, 10, 15});
    bool round_trip = true;
    for (size_t i = 0; i < g.get_total_cell_count(); ++i) round_trip &= g.get_cell_idx(g.get_cell_integer_coordinates(i)) == i;
    EXPECT(round_trip);
  }
  {   // serde.rs:228-374: round trips, files, and the V1 golden files
    using namespace mesh_to_sdf::serde;
    std::vector<Vec3> queries = {{1.f, 2.f, 3.f}, {6.f, 5.f, 4.f}};
    std::vector<float> distances = {1.0f, 3.0f};
    SerializeSdf<Vec3> ser = SerializeGeneric<Vec3>{queries.data(), queries.size(), distances.data(), distances.size()};
    auto data = serialize(ser);
    EXPECT(data == slurp(gold + "/sdf_generic_v1.bin"));
    auto de = deserialize<Vec3>(data);
    auto* dg = std::get_if<DeserializeGeneric<Vec3>>(&de);
    EXPECT(dg && dg->query_points == queries && dg->distances == distances);

    auto grid = Grid<P>::new_({1.f, 2.f, 3.f}, {4.f, 5.f, 6.f}, {7, 8, 9});
    std::vector<float> gd(grid.get_total_cell_count());
    for (size_t i = 0; i < gd.size(); ++i) gd[i] = (float)i;
    SerializeSdf<P> gser = SerializeGrid<P>{&grid, gd.data(), gd.size()};
    auto gdata = serialize(gser);
    EXPECT(gdata == slurp(gold + "/sdf_grid_v1.bin"));
    auto gde = deserialize<P>(gdata);
    auto* gg = std::get_if<DeserializeGrid<P>>(&gde);
    EXPECT(gg && gg->grid == grid && gg->distances == gd);

    const std::string path = tmp + "/m2s_cpp_sdf.bin";
    save_to_file(ser, path);
    EXPECT(slurp(path) == data);
    auto fde = read_from_file<Vec3>(path);
    auto* fg = std::get_if<DeserializeGeneric<Vec3>>(&fde);
    EXPECT(fg && fg->query_points == queries && fg->distances == distances);
    std::remove(path.c_str());

    auto old_generic = read_from_file<Vec3>(gold + "/sdf_generic_v1.bin");   // test_backward_compatibility_serde_generic_v1
    auto* og = std::get_if<DeserializeGeneric<Vec3>>(&old_generic);
    EXPECT(og && og->query_points == queries && og->distances == distances);
    auto old_grid = read_from_file<P>(gold + "/sdf_grid_v1.bin");             // test_backward_compatibility_serde_grid_v1
    auto* ogr = std::get_if<DeserializeGrid<P>>(&old_grid);
    EXPECT(ogr && ogr->grid == grid && ogr->distances == gd);

    bool io_error = false, de_error = false;
    try { read_from_file<P>(tmp + "/does/not/exist.bin"); } catch (const SerdeError& e) { io_error = e.kind == SerdeErrorKind::IoError; }
    try { std::vector<uint8_t> junk = {0x81, 0xa2, 'V', '2'}; deserialize<P>(junk); } catch (const SerdeError& e) { de_error = e.kind == SerdeErrorKind::DeserializationFailed; }
    EXPECT(io_error && de_error);
  }
  {   // panics of the reference
    std::vector<P> v = {{0.f, 1.f, 0.f}, {1.f, 2.f, 3.f}, {1.f, 3.f, 4.f}};
    std::vector<uint32_t> bad = {0, 1, 9};
    std::vector<P> q = {{0.f, 0.f, 0.f}};
    bool panicked = false;
    try { generate_sdf(v, Topology<uint32_t>::TriangleList(bad), q); } catch (const Panic&) { panicked = true; }
    EXPECT(panicked);
    std::vector<P> none;
    panicked = false;
    try { generate_sdf(none, Topology<uint32_t>::TriangleList(), q, AccelerationMethod::Rtree()); } catch (const Panic&) { panicked = true; }
    EXPECT(panicked);
    EXPECT(generate_sdf(none, Topology<uint32_t>::TriangleList(), q, AccelerationMethod::RtreeBvh()).empty());   // rtree_bvh.rs:104-106
  }
  std::printf("%d checks, %d failed\n", checks, fails);
  return fails ? 1 : 0;
}
