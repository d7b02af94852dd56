// pathtracer <scene_directory> [--device N | --gpus N | --devices a,b,...] [--partition tiles|samples] [--seed S]
//            [--flush-every K] [--resume 1]
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
//
// Several GPUs (--gpus N: devices 0 .. N-1; --devices: any list): the scene is built once, on the first GPU, and
// copied to the others device to device (ptk_scene_clone: no hierarchy is built twice), one host thread per GPU
// drives that GPU's own stream, and every iteration is dealt out either as interleaved bands of eight image rows
// (--partition tiles, the default: a GPU renders all samples of its own pixels) or as slices of the iteration's sample
// indices (--partition samples: every GPU renders every pixel).  This replaces the reference's pool of threads pulling
// pixel ranges from a job vector (app/pathtracer.cpp:337-365).  At every flush the per-GPU films are folded, on the
// first GPU, into one film (ptk_film_merge: device-to-device copies over xGMI and one kernel per film, the
// sample-count-weighted mean per pixel), and that one film is copied into the mapped files: one film crosses PCIe per
// flush, however many GPUs render.  The image does not depend on the number of GPUs or on the partition (tiles: bit for
// bit; samples: up to the rounding of the merged mean).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
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

struct Worker { // one GPU: its copy of the scene and its film
    int device = 0;
    PtkScene* scene = nullptr;
    PtkFilm* film = nullptr;
    int status = PTK_OK;
    std::string error;
};

// This worker's share of an iteration of `n_samples` samples starting at sample index `first_sample`.
void render_share(Worker& w, size_t rank, size_t world, bool tiles, int32_t width, int32_t height, int32_t first_sample, int32_t n_samples,
                  uint64_t seed) {
    PtkRenderJob job{};
    job.seed = seed;
    if (tiles && world > 1) { // bands of 8 rows, band b belongs to worker b % world
        const int64_t band_rows = 8, band_pixels = band_rows * width;
        const int64_t n_bands = (height + band_rows - 1) / band_rows;
        int64_t pixels = 0;
        for (int64_t band = static_cast<int64_t>(rank); band < n_bands; band += static_cast<int64_t>(world))
            pixels += std::min<int64_t>(band_rows, height - band * band_rows) * width;
        if (pixels == 0) return;
        job.start_idx = static_cast<int64_t>(rank) * band_pixels;
        job.number_of_pixels = pixels;
        job.number_of_samples = n_samples;
        job.sample_offset = first_sample;
        w.status = ptk_render_bands(w.scene, w.film, &job, band_pixels, static_cast<int64_t>(world) * band_pixels);
    }
    else { // a contiguous slice of the iteration's sample indices
        const int32_t base = n_samples / static_cast<int32_t>(world), extra = n_samples % static_cast<int32_t>(world);
        const int32_t r = static_cast<int32_t>(rank);
        const int32_t count = base + (r < extra ? 1 : 0), first = r * base + std::min(r, extra);
        if (count == 0) return;
        job.start_idx = 0;
        job.number_of_pixels = static_cast<int64_t>(width) * height;
        job.number_of_samples = count;
        job.sample_offset = first_sample + first;
        w.status = ptk_render(w.scene, w.film, &job);
    }
    if (w.status != PTK_OK) w.error = ptk_last_error(); // the message is thread local
}
} // namespace

int main(int argc, char* argv[]) {
    if (argc < 2 || argv[1][0] == '-') {
        std::cerr << "Invalid arguments provided.\n"
                     "Usage: pathtracer <scene_directory> [--device N | --gpus N | --devices a,b,...] [--partition tiles|samples]\n"
                     "                  [--seed S] [--flush-every K] [--resume 1]\n\n"
                     "positional arguments:\n"
                     "   scene_directory                  path to the scene directory (relative to main project directory)\n";
        return EXIT_FAILURE;
    }
    const std::string dir = argv[1];
    int flush_every = 0, resume = 0;
    unsigned long long seed = 0;
    std::vector<int> devices;
    bool tiles = true;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--device")) devices.assign(1, std::atoi(argv[i + 1]));
        else if (!std::strcmp(argv[i], "--gpus")) {
            devices.clear();
            for (int k = 0; k < std::atoi(argv[i + 1]); k++) devices.push_back(k);
        }
        else if (!std::strcmp(argv[i], "--devices")) {
            devices.clear();
            for (const char* p = argv[i + 1]; *p;) {
                char* end = nullptr;
                devices.push_back(static_cast<int>(std::strtol(p, &end, 10)));
                if (end == p) return die("--devices: expected a comma-separated list of device indices");
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (!std::strcmp(argv[i], "--partition")) {
            if (std::strcmp(argv[i + 1], "tiles") && std::strcmp(argv[i + 1], "samples")) return die("--partition: tiles or samples");
            tiles = !std::strcmp(argv[i + 1], "tiles");
        }
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
    if (devices.empty()) devices.push_back(0);
    std::vector<Worker> workers(devices.size());
    // the scene is built once, on the first GPU ...
    workers[0].device = devices[0];
    if (ptk_init(devices[0]) != PTK_OK) return die(ptk_last_error());
    if (ptk_scene_create(ptk_host_scene_desc(loaded), 0, &workers[0].scene) != PTK_OK) return die(ptk_last_error());
    if (ptk_film_create(settings.width, settings.height, &workers[0].film) != PTK_OK) return die(ptk_last_error());
    ptk_host_free_scene(loaded);
    // ... and copied to the others, device to device, all at once
    {
        std::vector<std::thread> threads;
        for (size_t k = 1; k < workers.size(); k++) {
            workers[k].device = devices[k];
            threads.emplace_back([&workers, &settings, k] {
                Worker& w = workers[k];
                if ((w.status = ptk_init(w.device)) == PTK_OK && (w.status = ptk_scene_clone(workers[0].scene, &w.scene)) == PTK_OK)
                    w.status = ptk_film_create(settings.width, settings.height, &w.film);
                if (w.status != PTK_OK) w.error = ptk_last_error();
            });
        }
        for (std::thread& t : threads) t.join();
        for (const Worker& w : workers)
            if (w.status != PTK_OK) return die(w.error);
    }
    // several GPUs: the film their films are folded into at every flush, on the first GPU
    PtkFilm* assembled = nullptr;
    if (workers.size() > 1) {
        if (ptk_init(devices[0]) != PTK_OK) return die(ptk_last_error());
        if (ptk_film_create(settings.width, settings.height, &assembled) != PTK_OK) return die(ptk_last_error());
    }
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

    // a resumed render continues in the first GPU's film; the others start empty and the merge adds them up
    if (resume && ptk_film_upload(workers[0].film, image, position, normal, counters.data()) != PTK_OK) return die(ptk_last_error());

    auto begin = std::chrono::steady_clock::now();
    auto last_flush = begin;
    print_progress(0);
    const int spp = settings.samples_per_pixel, spi = settings.samples_per_iteration;
    const int first_sample = progress.samples_done;
    const int iterations = spp > first_sample ? (spp - first_sample - 1) / spi + 1 : 0;
    for (int j = 0; j < iterations; j++) {
        const int32_t n_samples = std::min(spi, spp - progress.samples_done);
        if (workers.size() == 1) render_share(workers[0], 0, 1, tiles, settings.width, settings.height, progress.samples_done, n_samples, seed);
        else { // one host thread per GPU
            std::vector<std::thread> threads;
            for (size_t k = 0; k < workers.size(); k++)
                threads.emplace_back(render_share, std::ref(workers[k]), k, workers.size(), tiles, settings.width, settings.height,
                                     progress.samples_done, n_samples, seed);
            for (std::thread& t : threads) t.join();
        }
        for (const Worker& w : workers)
            if (w.status != PTK_OK) return die(w.error);
        progress.samples_done += n_samples;
        auto now = std::chrono::steady_clock::now();
        bool flush = j + 1 == iterations;
        if (flush_every > 0) flush |= (j + 1) % flush_every == 0;
        else flush |= std::chrono::duration_cast<std::chrono::milliseconds>(now - last_flush).count() >= 1000;
        if (flush) {
            PtkFilm* whole = workers[0].film;
            if (assembled) {
                // Every GPU's film folded into one on the first GPU, in the order of the GPUs: a pixel only one GPU has
                // rendered keeps its bits, the others get increment_average's arithmetic (ptk_film_merge).  The films
                // themselves stay as they are and go on accumulating.
                if (ptk_film_clear(assembled) != PTK_OK) return die(ptk_last_error());
                for (Worker& w : workers)
                    if (ptk_film_merge(assembled, w.film) != PTK_OK) return die(ptk_last_error());
                whole = assembled;
            }
            if (ptk_film_download(whole, image, position, normal, counters.data()) != PTK_OK) return die(ptk_last_error());
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
    uint64_t paths = 0, rays = 0;
    for (const Worker& w : workers) {
        PtkStats st;
        if (ptk_get_stats(w.scene, &st) != PTK_OK) continue;
        paths += st.paths;
        rays += st.extension_rays + st.shadow_rays;
    }
    if (seconds > 0)
        std::clog << paths << " paths, " << rays << " rays, " << paths / seconds / 1e6 << " Mpaths/s on " << workers.size() << " GPU(s)" << std::endl;

    ptk_host_unmap_buffer(image, file_size, image_fd);
    ptk_host_unmap_buffer(position, file_size, position_fd);
    ptk_host_unmap_buffer(normal, file_size, normal_fd);
    ptk_film_destroy(assembled);
    for (Worker& w : workers) {
        ptk_film_destroy(w.film);
        ptk_scene_destroy(w.scene);
    }
    return 0;
}
