// snapshot_info.cpp -- host-only tool: parse a snapshot with the C++ host mirror's own code path
// (msgpack_lite + NerfRender::load_snapshot/reset_network logic) WITHOUT touching a GPU and print
// the derived model description.  Used by the CPU tests to check the C++ mirror against the Python one.
#include <cstdio>
#include <string>

#include "nerf_render.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  try {
    ngp::NerfRender r(-1);  // host-only: no device context is created
    r.load_snapshot(argv[1]);
    r.reset_network();
    const nrf_model_desc& d = r.model_desc();
    uint64_t expect = 0;
    const int rc = nrf_expected_n_params(&d, &expect);
    double psum = 0.0, gsum = 0.0;  // order-dependent checksums of the arrays as parsed
    for (uint64_t i = 0; i < d.n_params; ++i) psum += (double)d.params[i] * (double)((i % 97) + 1);
    for (uint64_t i = 0; i < d.n_density_grid; ++i) gsum += (double)d.density_grid[i] * (double)((i % 89) + 1);
    std::printf("{\"psum\": %.17g, \"gsum\": %.17g, \"grid_type\": %u, \"n_levels\": %u, \"F\": %u, \"log2T\": %u, \"base\": %u, \"pls\": %.9g, "
                "\"n_neurons\": %u, \"dh\": %u, \"da\": %u, \"doa\": %u, \"dno\": %u, \"sa\": %u, \"rh\": %u, \"ra\": %u, "
                "\"roa\": %u, \"dir\": %u, \"shdeg\": %u, \"nfreq\": %u, \"bound\": %.9g, \"scale\": %.9g, \"cascade\": %u, "
                "\"H\": %u, \"mean_density\": %.9g, \"n_params\": %llu, \"n_grid\": %llu, \"expected\": %llu, \"rc\": %d}\n",
                psum, gsum, d.grid_type, d.n_levels, d.n_features_per_level, d.log2_hashmap_size, d.base_resolution, d.per_level_scale,
                d.n_neurons, d.density_hidden_layers, d.density_activation, d.density_output_activation, d.density_n_output,
                d.sigma_activation, d.rgb_hidden_layers, d.rgb_activation, d.rgb_output_activation, d.dir_encoding,
                d.sh_degree, d.n_frequencies, d.bound, d.scale, d.cascade, d.density_grid_size, d.mean_density,
                (unsigned long long)d.n_params, (unsigned long long)d.n_density_grid, (unsigned long long)expect, rc);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
