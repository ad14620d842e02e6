// TEST INFRASTRUCTURE.  Link-time stand-ins for the symbols of the reference's custom_rasterizer that live in its CUDA / hierarchy
// translation units (rasterizer_gpu.cu, grid_neighbor.cpp), so that its CPU rasterizer (rasterizer.cpp:94-133, compiled from where it
// lies under /root/reference by oracle/ref_build.py) loads as a Python extension on its own.  Nothing here restates reference code.
#include "rasterizer.h"
#include <stdexcept>

std::vector<torch::Tensor> rasterize_image_gpu(torch::Tensor, torch::Tensor, torch::Tensor, int, int, float, int) {
    throw std::runtime_error("oracle/_ref holds only the CPU half of custom_rasterizer");
}
std::vector<std::vector<torch::Tensor>> build_hierarchy(std::vector<torch::Tensor>, std::vector<torch::Tensor>, int, int) {
    throw std::runtime_error("not built in oracle/_ref");
}
std::vector<std::vector<torch::Tensor>> build_hierarchy_with_feat(std::vector<torch::Tensor>, std::vector<torch::Tensor>,
                                                                  std::vector<torch::Tensor>, int, int) {
    throw std::runtime_error("not built in oracle/_ref");
}
