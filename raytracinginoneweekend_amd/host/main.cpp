// Drop-in replacement of the reference executable (reference: main.cpp:20-81): same positional
// arguments `width height spp depth`, same scene, same camera, same output file `image_binary.ppm`
// in P6 format, same progress/timing lines on stderr — with the "Image Generation" scope executed
// on an MI355X through librtk.so instead of the CPU loops.
//
// Extra, optional: a 5th argument is the RNG seed (default 0), a 6th the HIP device index — or a comma-separated LIST of
// devices ("0,1,2,3,4,5,6,7"): the image is then sharded by interleaved tiles over those GPUs from this one process and
// gathered over RCCL (rtk_multi_*; same bytes as on one device) —, a 7th the number of
// progressive passes the samples are split into (default 1; progress is reported after each pass), an 8th selects how
// the closest hit is found: 0 = the reference's list walk (default), 1 = uniform grid (same image, fewer sphere tests).
// A 9th argument names a checkpoint file: with more than one pass the running sums are saved there after every pass and a
// later run with the same arguments resumes after the last completed pass (a 10th argument stops this run after that
// many passes: exit status 3, no image yet).
#include "MathUtils.hpp"

#include "Camera.hpp"
#include "Color.hpp"
#include "HittableList.hpp"
#include "Material.hpp"
#include "ProfileLogScope.hpp"
#include "Render.hpp"
#include "Sphere3.hpp"

#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    // Image (main.cpp:23-41)
    float aspect_ratio = 3.0f / 2.0f;
    const int image_width = argc > 1 ? static_cast<int>(std::stoll(argv[1])) : 400;
    int image_height;
    if (argc > 2) {
        image_height = static_cast<int>(std::stoll(argv[2]));
        aspect_ratio = image_width / static_cast<float>(image_height);
    } else {
        image_height = static_cast<int>(image_width / aspect_ratio);
    }
    const int samples_per_pixel = argc > 3 ? static_cast<int>(std::stoll(argv[3])) : 100;
    const int max_depth = argc > 4 ? static_cast<int>(std::stoll(argv[4])) : 50;
    const uint64_t seed = argc > 5 ? std::stoull(argv[5]) : 0;
    std::vector<int32_t> devices;
    if (argc > 6) {
        const std::string list = argv[6];
        for (size_t pos = 0; pos <= list.size();) {
            const size_t comma = list.find(',', pos);
            devices.push_back(static_cast<int32_t>(std::stoll(list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos))));
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
    }
    if (devices.empty()) devices.push_back(0);
    const int device = devices[0];
    const int batches = argc > 7 ? static_cast<int>(std::stoll(argv[7])) : 1;
    const int accel = argc > 8 ? static_cast<int>(std::stoll(argv[8])) : RTK_ACCEL_LIST;
    const std::string checkpoint = argc > 9 ? argv[9] : "";
    const int max_passes = argc > 10 ? static_cast<int>(std::stoll(argv[10])) : 0;

    // World and camera (main.cpp:45-54)
    HittableList world = random_scene();
    Camera camera{Point3{13.0f, 2.0f, 3.0f}, Point3{0.0f, 0.0f, 0.0f}, Vector3{0.0f, 1.0f, 0.0f}, 20, aspect_ratio, 0.1f, 10.0f};

    if (devices.size() > 1 && (batches > 1 || !checkpoint.empty() || max_passes > 0)) {
        std::cerr << "error: progressive passes, checkpoints and max_passes are single-device options; a device list renders the frame in one pass\n";
        return 2;
    }
    // The file is opened only when the frame is complete: the reference truncates image_binary.ppm before rendering
    // (main.cpp:59) and leaves a partial file when interrupted; a run that stops early here (exit status 3, an error)
    // leaves the previous file untouched.
    std::ostringstream frame_bytes(std::ios_base::binary);
    std::ostream& bin_file = frame_bytes;
    try {
        rtk_stats stats{};
        {
            PROFILE_LOG_SCOPE("Image Generation");
            std::cerr << "\rScanlines remaining: " << image_height - 1 << ' ' << std::flush;
            if (devices.size() > 1) {
                stats = rtiow::render_ppm_multi(bin_file, world, camera, image_width, image_height, samples_per_pixel, max_depth, seed, devices, accel);
            } else if (batches <= 1) {
                // one pass: gamma and quantisation run on the device, the file is one write
                stats = rtiow::render_ppm(bin_file, world, camera, image_width, image_height, samples_per_pixel, max_depth, seed, device, accel);
            } else {
                const rtiow::RenderResult frame = rtiow::render_image_progressive(
                    world, camera, image_width, image_height, samples_per_pixel, max_depth, batches,
                    [](int done, int total) { std::cerr << "\rSamples remaining: " << total - done << ' ' << std::flush; }, seed, device, accel, checkpoint,
                    max_passes);
                if (frame.passes_done < batches) {
                    std::cerr << "\nstopped after pass " << frame.passes_done << " of " << batches << "; resume with the same arguments\n";
                    return 3;
                }
                rtiow::write_ppm_binary(bin_file, image_width, image_height, frame.rgb_sum.data(), samples_per_pixel);
                stats = frame.stats;
            }
            std::cerr << "\rScanlines remaining: 0 \nDone.\n";
        }
        std::cerr << "rays " << stats.rays << "  kernel " << stats.kernel_ms << " ms  "
                  << (stats.kernel_ms > 0 ? stats.rays / (stats.kernel_ms * 1e3) : 0.0) << " Mrays/s\n";
    } catch (const std::exception& e) {
        std::cerr << "\nerror: " << e.what() << '\n';
        return 1;
    }
    std::ofstream out_file("image_binary.ppm", std::ios_base::binary);
    const std::string payload = frame_bytes.str();
    out_file.write(payload.data(), static_cast<std::streamsize>(payload.size()));
    out_file.close();
    if (!out_file) {
        std::cerr << "error: cannot write image_binary.ppm\n";
        return 1;
    }
    return 0;
}
