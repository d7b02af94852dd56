// denoiser <scene_directory> [--device N]
//
// Command-line driver with the reference's contract (app/denoiser.cpp:10-54): reads settings.json and
// denoising.json, maps ./temp/raw_pixel.dat, raw_position.dat and raw_normal.dat as they are, and
// writes the filtered image to ./temp/raw_denoised.dat.  The filters run on the GPU.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "ptk.h"
#include "ptk_host.h"

namespace {
int die(const std::string& message) {
    std::cerr << "denoiser: " << message << std::endl;
    return EXIT_FAILURE;
}
} // namespace

int main(int argc, char* argv[]) {
    if (argc < 2 || argv[1][0] == '-') {
        std::cerr << "Invalid arguments provided.\n"
                     "Usage: denoiser <scene_directory> [--device N]\n\n"
                     "positional arguments:\n"
                     "   scene_directory                  path to the scene directory (relative to main project directory)\n";
        return EXIT_FAILURE;
    }
    const std::string dir = argv[1];
    int device = 0;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--device")) device = std::atoi(argv[i + 1]);
        else return die(std::string("unknown option ") + argv[i]);
    }
    PtkSettings settings;
    if (ptk_host_load_settings((dir + "/settings.json").c_str(), &settings) != PTK_OK) return die(ptk_host_last_error());
    PtkDenoiseTask* tasks = nullptr;
    int32_t n_tasks = 0;
    if (ptk_host_load_denoising((dir + "/denoising.json").c_str(), &settings, &tasks, &n_tasks) != PTK_OK)
        return die(ptk_host_last_error());
    if (n_tasks == 0) {
        std::clog << "No denoising mode set in denoising settings, exiting..." << std::endl;
        return 0;
    }
    const size_t file_size = static_cast<size_t>(settings.width) * static_cast<size_t>(settings.height) * 3 * sizeof(double);
    int image_fd, position_fd, normal_fd, denoised_fd;
    double* image = ptk_host_map_buffer("./temp/raw_pixel.dat", file_size, 0, &image_fd);
    double* position = ptk_host_map_buffer("./temp/raw_position.dat", file_size, 0, &position_fd);
    double* normal = ptk_host_map_buffer("./temp/raw_normal.dat", file_size, 0, &normal_fd);
    if (!image || !position || !normal) return die(ptk_host_last_error());
    std::clog << "Denoising..." << std::endl;
    double* denoised = ptk_host_map_buffer("./temp/raw_denoised.dat", file_size, 1, &denoised_fd);
    if (!denoised) return die(ptk_host_last_error());
    std::memcpy(denoised, image, file_size);
    if (ptk_init(device) != PTK_OK) return die(ptk_last_error());
    if (ptk_denoise(tasks, n_tasks, settings.width, settings.height, denoised, position, normal) != PTK_OK)
        return die(ptk_last_error());
    std::free(tasks);
    ptk_host_unmap_buffer(denoised, file_size, denoised_fd);
    ptk_host_unmap_buffer(image, file_size, image_fd);
    ptk_host_unmap_buffer(position, file_size, position_fd);
    ptk_host_unmap_buffer(normal, file_size, normal_fd);
    return 0;
}
