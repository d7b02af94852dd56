// pathtracer <scene_directory> [--device N] [--seed S] [--flush-every K] [--resume 1]
//
// Command-line driver with the reference's contract (app/pathtracer.cpp:300-380): reads
// <scene_directory>/settings.json and scene.json, renders samples_per_pixel samples in iterations of
// samples_per_iteration, and keeps the running means in the shared mappings ./temp/raw_pixel.dat,
// ./temp/raw_position.dat and ./temp/raw_normal.dat (float64 [H][W][3]) that `denoiser` and
// python_utils/to_png.py read.  The reference's thread pool and job vector are replaced by one job
// per iteration covering the whole frame on the GPU; the files are refreshed every K iterations
// (default: about once a second), so a viewer can still snapshot a render in progress.  With every refresh
// the driver also writes ./temp/raw_progress.dat (per-pixel sample counts, the seed and the number of samples
// done); `--resume 1` picks a render up from those files and continues to samples_per_pixel, drawing the sample
// indices that are still missing (the reference cannot resume: its random streams are not addressable).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "ptk.h"
#include "ptk_host.h"

namespace {
void print_progress(int progress) { // same bar as app/pathtracer.cpp:203-219
    const int bar_width = 60;
    int pos = static_cast<int>(bar_width * progress / 100.0);
    std::clog << "[";
    for (int i = 0; i < bar_width; ++i) std::clog << (i < pos ? "=" : (i == pos ? ">" : " "));
    std::clog << "] " << progress << " %\r";
}

struct ProgressHeader { // ./temp/raw_progress.dat: this header, then uint64 sample counts [H][W]
    char magic[8];
    int32_t width, height;
    uint64_t seed;
    int32_t samples_done, reserved;
};
const char PROGRESS_MAGIC[8] = {'P', 'T', 'K', 'P', 'R', 'O', 'G', '1'};
const char* PROGRESS_FILE = "./temp/raw_progress.dat";

bool write_progress(const ProgressHeader& header, const std::vector<uint64_t>& counters) {
    const std::string tmp = std::string(PROGRESS_FILE) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(&header, sizeof(header), 1, f) == 1 &&
              std::fwrite(counters.data(), sizeof(uint64_t), counters.size(), f) == counters.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok && std::rename(tmp.c_str(), PROGRESS_FILE) == 0; // readers never see a half-written file
}

bool read_progress(ProgressHeader& header, std::vector<uint64_t>& counters) {
    FILE* f = std::fopen(PROGRESS_FILE, "rb");
    if (!f) return false;
    bool ok = std::fread(&header, sizeof(header), 1, f) == 1 && std::memcmp(header.magic, PROGRESS_MAGIC, 8) == 0 &&
              std::fread(counters.data(), sizeof(uint64_t), counters.size(), f) == counters.size();
    std::fclose(f);
    return ok;
}

int die(const std::string& message) {
    std::cerr << "pathtracer: " << message << std::endl;
    return EXIT_FAILURE;
}
} // namespace

int main(int argc, char* argv[]) {
    if (argc < 2 || argv[1][0] == '-') {
        std::cerr << "Invalid arguments provided.\n"
                     "Usage: pathtracer <scene_directory> [--device N] [--seed S] [--flush-every K] [--resume 1]\n\n"
                     "positional arguments:\n"
                     "   scene_directory                  path to the scene directory (relative to main project directory)\n";
        return EXIT_FAILURE;
    }
    const std::string dir = argv[1];
    int device = 0, flush_every = 0, resume = 0;
    unsigned long long seed = 0;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--device")) device = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--seed")) seed = std::strtoull(argv[i + 1], nullptr, 10);
        else if (!std::strcmp(argv[i], "--flush-every")) flush_every = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--resume")) resume = std::atoi(argv[i + 1]);
        else return die(std::string("unknown option ") + argv[i]);
    }

    auto begin_build = std::chrono::steady_clock::now();
    PtkSettings settings;
    if (ptk_host_load_settings((dir + "/settings.json").c_str(), &settings) != PTK_OK) return die(ptk_host_last_error());
    PtkLoadedScene* loaded = nullptr;
    if (ptk_host_load_scene((dir + "/scene.json").c_str(), &settings, dir.c_str(), &loaded) != PTK_OK)
        return die(ptk_host_last_error());
    if (ptk_init(device) != PTK_OK) return die(ptk_last_error());
    PtkScene* scene = nullptr;
    if (ptk_scene_create(ptk_host_scene_desc(loaded), 0, &scene) != PTK_OK) return die(ptk_last_error());
    ptk_host_free_scene(loaded);
    auto end_build = std::chrono::steady_clock::now();
    std::clog << "Time taken to build scene: " << std::chrono::duration_cast<std::chrono::seconds>(end_build - begin_build).count()
              << "[s]" << std::endl;

    const size_t pixel_count = static_cast<size_t>(settings.width) * static_cast<size_t>(settings.height);
    const size_t file_size = pixel_count * 3 * sizeof(double);
    int image_fd, position_fd, normal_fd;
    std::vector<uint64_t> counters(pixel_count, 0);
    ProgressHeader progress{};
    std::memcpy(progress.magic, PROGRESS_MAGIC, 8);
    progress.width = settings.width;
    progress.height = settings.height;
    progress.seed = seed;
    if (resume) {
        ProgressHeader found{};
        if (!read_progress(found, counters)) return die(std::string("--resume: cannot read ") + PROGRESS_FILE);
        if (found.width != settings.width || found.height != settings.height || found.seed != seed)
            return die("--resume: the saved render has another size or seed");
        progress.samples_done = found.samples_done;
    }
    double* image = ptk_host_map_buffer("./temp/raw_pixel.dat", file_size, resume ? 0 : 1, &image_fd);
    double* position = ptk_host_map_buffer("./temp/raw_position.dat", file_size, resume ? 0 : 1, &position_fd);
    double* normal = ptk_host_map_buffer("./temp/raw_normal.dat", file_size, resume ? 0 : 1, &normal_fd);
    if (!image || !position || !normal) return die(ptk_host_last_error());

    PtkFilm* film = nullptr;
    if (ptk_film_create(settings.width, settings.height, &film) != PTK_OK) return die(ptk_last_error());
    if (resume && ptk_film_upload(film, image, position, normal, counters.data()) != PTK_OK) return die(ptk_last_error());

    auto begin = std::chrono::steady_clock::now();
    auto last_flush = begin;
    print_progress(0);
    const int spp = settings.samples_per_pixel, spi = settings.samples_per_iteration;
    const int first_sample = progress.samples_done;
    const int iterations = spp > first_sample ? (spp - first_sample - 1) / spi + 1 : 0;
    for (int j = 0; j < iterations; j++) {
        PtkRenderJob job{};
        job.start_idx = 0;
        job.number_of_pixels = static_cast<int64_t>(pixel_count);
        job.number_of_samples = std::min(spi, spp - progress.samples_done);
        job.sample_offset = progress.samples_done;
        job.seed = seed;
        if (ptk_render(scene, film, &job) != PTK_OK) return die(ptk_last_error());
        progress.samples_done += job.number_of_samples;
        auto now = std::chrono::steady_clock::now();
        bool flush = j + 1 == iterations;
        if (flush_every > 0) flush |= (j + 1) % flush_every == 0;
        else flush |= std::chrono::duration_cast<std::chrono::milliseconds>(now - last_flush).count() >= 1000;
        if (flush) {
            if (ptk_film_download(film, image, position, normal, counters.data()) != PTK_OK) return die(ptk_last_error());
            if (!write_progress(progress, counters)) return die(std::string("cannot write ") + PROGRESS_FILE);
            last_flush = std::chrono::steady_clock::now();
            print_progress(static_cast<int>(100.0 * (j + 1) / iterations));
        }
    }
    print_progress(100);
    std::clog << std::endl;
    auto end = std::chrono::steady_clock::now();
    const double seconds = std::chrono::duration_cast<std::chrono::milliseconds>(end - begin).count() / 1e3;
    std::clog << "Program complete. Time taken: " << std::chrono::duration_cast<std::chrono::seconds>(end - begin).count() << "[s]"
              << std::endl;
    PtkStats st;
    if (ptk_get_stats(scene, &st) == PTK_OK && seconds > 0)
        std::clog << st.paths << " paths, " << (st.extension_rays + st.shadow_rays) << " rays, " << st.paths / seconds / 1e6
                  << " Mpaths/s" << std::endl;

    ptk_host_unmap_buffer(image, file_size, image_fd);
    ptk_host_unmap_buffer(position, file_size, position_fd);
    ptk_host_unmap_buffer(normal, file_size, normal_fd);
    ptk_film_destroy(film);
    ptk_scene_destroy(scene);
    return 0;
}
