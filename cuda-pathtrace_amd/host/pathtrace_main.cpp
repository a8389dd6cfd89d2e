// pathtrace_main.cpp -- command-line front end equivalent to the single-frame mode of
// src/main.cu:18-83,179-193, written against the look-alike headers exactly as main.cu is
// written against the reference's (Scene / Renderer / Camera / OutputBuffer).
// Same flags, defaults, banner and timing line; -i/-d (OpenGL window, embedded-Python CNN
// denoiser) are accepted and reported as unsupported: both are out of scope.
// Additions: --rng, --max-bounces, --spheres N (seeded random scene), --frames N (headless
// stand-in for the interactive loop main.cu:146-177: N x Render() into the same device buffer),
// --poses FILE (scripted fly-through: one "x y z yaw pitch" line per frame, the pose-list format of
// collect_data.py:20-31; generator state carries over from frame to frame like the reference's
// interactive mode, per-frame times are summarised), --gpus N (the frame row-tiled over devices
// --device .. --device+N-1 by one process: MultiRenderer / pt_mgpu_*, RCCL gather to the first device).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "Camera.h"
#include "Denoiser.h"
#include "MultiRenderer.h"
#include "OutputBuffer.h"
#include "Renderer.h"
#include "Scene.h"

static void usage() {
  std::cout << "cuda-pathtrace" << std::endl
            << "Options:\n"
               "  -h [ --help ]                 Print help messages\n"
               "  -t [ --threads-per-block ] arg Number of threads per block in 2D CUDA scheduling grid. (ignored)\n"
               "  --size arg                    Size of the screen in pixels\n"
               "  -s [ --samples ] arg          Number of samples per pixel\n"
               "  --device arg                  Which device to use for rendering\n"
               "  -d [ --denoising ]            Use denoising neural network. (unsupported)\n"
               "  -i [ --interactive ]          Interactive mode (unsupported; see --frames)\n"
               "  --nobitmap                    Don't output bitmaps for each channel\n"
               "  -o [ --output ] arg           Prefix of output file/path\n"
               "  -x [ --camera-x ] arg         Starting camera position x\n"
               "  -y [ --camera-y ] arg         Starting camera position y\n"
               "  -z [ --camera-z ] arg         Starting camera position z\n"
               "  -c [ --camera-yaw ] arg       Starting camera view yaw\n"
               "  -p [ --camera-pitch ] arg     Starting camera view pitch\n"
               "  --rng arg                     xorwow (default, reference parity) | philox\n"
               "  --max-bounces arg             path length cap (default 5)\n"
               "  --spheres arg                 render a seeded random scene of N spheres\n"
               "  --frames arg                  render N frames back to back (headless interactive loop)\n"
               "  --poses arg                   fly-through: file with one 'x y z yaw pitch' line per frame\n"
               "  --batch                       with --poses: render the fly-through in batches of 32 frames per launch\n"
               "  --gpus arg                    row-tile the frame over N devices starting at --device (RCCL gather)\n"
               "  --preview arg                 also write the display-packed frame (Denoiser) as a binary PPM\n"
            << std::endl;
}

int main(int argc, const char** argv) {
  // default arguments (main.cu:20-29)
  int size = 512;
  int threadsPerBlock = 8;
  int samplesPerPixel = 4;
  int cudaDevice = 0;
  float cameraPos[3] = {50.0f, 52.0f, 295.6f};
  float cameraView[2] = {-90.0f, 0.0f};
  bool denoising = false, interactive = false, noBitmap = false;
  std::string outputName = "output/out";
  std::string rng = "xorwow";
  int maxBounces = 5, nSpheres = 0, frames = 1, gpus = 1;
  std::string posesFile, previewFile;
  bool batch = false;        // --poses: all frames in one call (one launch per 32 frames)
  void* batch_frames = NULL;

  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto value = [&](const char* what) -> const char* {
      if (i + 1 >= argc) {
        std::cerr << "ERROR: the required argument for option '" << what << "' is missing" << std::endl << std::endl;
        usage();
        exit(1);
      }
      return argv[++i];
    };
    if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (a == "-t" || a == "--threads-per-block") threadsPerBlock = atoi(value("--threads-per-block"));
    else if (a == "--size") size = atoi(value("--size"));
    else if (a == "-s" || a == "--samples") samplesPerPixel = atoi(value("--samples"));
    else if (a == "--device") cudaDevice = atoi(value("--device"));
    else if (a == "-d" || a == "--denoising") denoising = true;
    else if (a == "-i" || a == "--interactive") interactive = true;
    else if (a == "--nobitmap") noBitmap = true;
    else if (a == "-o" || a == "--output") outputName = value("--output");
    else if (a == "-x" || a == "--camera-x") cameraPos[0] = (float)atof(value("--camera-x"));
    else if (a == "-y" || a == "--camera-y") cameraPos[1] = (float)atof(value("--camera-y"));
    else if (a == "-z" || a == "--camera-z") cameraPos[2] = (float)atof(value("--camera-z"));
    else if (a == "-c" || a == "--camera-yaw") cameraView[0] = (float)atof(value("--camera-yaw"));
    else if (a == "-p" || a == "--camera-pitch") cameraView[1] = (float)atof(value("--camera-pitch"));
    else if (a == "--rng") rng = value("--rng");
    else if (a == "--max-bounces") maxBounces = atoi(value("--max-bounces"));
    else if (a == "--spheres") nSpheres = atoi(value("--spheres"));
    else if (a == "--frames") frames = atoi(value("--frames"));
    else if (a == "--gpus") gpus = atoi(value("--gpus"));
    else if (a == "--poses") posesFile = value("--poses");
    else if (a == "--batch") batch = true;
    else if (a == "--preview") previewFile = value("--preview");
    else {
      std::cerr << "ERROR: unrecognised option '" << a << "'" << std::endl << std::endl;
      usage();
      return 1;
    }
  }
  int width, height;
  width = height = size;  // main.cu:66-67
  std::cout << "cuda-pathtrace 0.3" << std::endl;
  std::cout << "------------------" << std::endl;
  std::cout << "Dimensions: " << width << " x " << height << std::endl;
  std::cout << "Threads per block: " << threadsPerBlock << std::endl;
  std::cout << "Samples per pixel: " << samplesPerPixel << std::endl;
  std::cout << "Using CUDA device: " << cudaDevice << std::endl;
  const bool multi = gpus > 1 || (getenv("PT_FORCE_MGPU") && atoi(getenv("PT_FORCE_MGPU")) != 0);
  if (gpus < 1) {
    std::cerr << "ERROR: --gpus must be at least 1" << std::endl;
    return 1;
  }
  if (multi) std::cout << "Row-tiled over " << gpus << " device(s) starting at " << cudaDevice << std::endl;
  if (!interactive)
    std::cout << "Output file prefix: " << outputName << std::endl;
  else
    std::cout << "Running in interactive mode: " << (denoising ? "denoising is on" : "denoising is off") << std::endl;
  std::cout << "Camera: " << cameraPos[0] << " " << cameraPos[1] << " " << cameraPos[2] << " " << cameraView[0] << " "
            << cameraView[1] << std::endl;
  if (interactive || denoising) {
    std::cerr << "ERROR: -i/-d need the OpenGL window / embedded-Python denoiser of the reference, which this build "
                 "does not include; use --frames N for a headless frame loop" << std::endl;
    return 1;
  }

  // set device (main.cu:86)
  gpuErrchk(pt_set_device(cudaDevice));

  // load scene and create renderer (main.cu:125-128)
  Scene scene = nSpheres > 0 ? Scene::Random(nSpheres, 1, true) : Scene();
  pt_renderer_opts opts;
  pt_renderer_opts_default(&opts);
  opts.max_bounces = maxBounces;
  opts.rng_mode = rng == "philox" ? PT_RNG_PHILOX : PT_RNG_XORWOW;
  // one device: the reference's Renderer; several: the same interface over pt_mgpu_* (frame and scene stay on
  // the first device, where main.cu has them)
  Renderer* single = NULL;
  MultiRenderer* tiled = NULL;
  if (multi) {
    std::vector<int> devices;
    for (int g = 0; g < gpus; g++) devices.push_back(cudaDevice + g);
    tiled = new MultiRenderer(devices, width, height, samplesPerPixel, threadsPerBlock, &opts);
    std::cout << "Exchange: " << tiled->Backend() << std::endl;
  } else {
    single = new Renderer(width, height, samplesPerPixel, threadsPerBlock, opts);
  }
  auto render = [&](OutputBuffer b, const Scene& s, const Camera& c) { return tiled ? tiled->Render(b, s, c) : single->Render(b, s, c); };
  Camera camera(glm::vec3(cameraPos[0], cameraPos[1], cameraPos[2]), cameraView[0], cameraView[1]);  // main.cu:128 verbatim

  // allocate output buffer (main.cu:131-139)
  OutputBuffer d_buffer(width, height);
  d_buffer.AllocateGPU();

  // render frame(s) (main.cu:182-183; --frames repeats the loop body of main.cu:146-148)
  float renderTime = 0.0f;
  if (!posesFile.empty()) {
    // headless version of the interactive loop: the camera moves, the same device buffer and the
    // same renderer (generator state included) are reused every frame (main.cu:146-148)
    std::ifstream in(posesFile.c_str());
    if (!in) {
      std::cerr << "ERROR: cannot open pose file " << posesFile << std::endl;
      return 1;
    }
    std::vector<float> times;
    std::vector<Camera> cameras;
    std::string line;
    while (std::getline(in, line)) {
      std::istringstream ls(line);
      float x, y, z, yaw, pitch;
      if (!(ls >> x >> y >> z >> yaw >> pitch)) continue;
      cameras.push_back(Camera(glm::vec3(x, y, z), yaw, pitch));
    }
    if (cameras.empty()) {
      std::cerr << "ERROR: no poses in " << posesFile << std::endl;
      return 1;
    }
    if (batch && single) {
      // every pose is known before the first frame: one call, one launch per 32 frames (Renderer::RenderFrames); each frame
      // has its own buffer, the last one is what gets saved
      void* d_frames = NULL;
      const size_t frame_floats = (size_t)width * height * 14;
      gpuErrchk(pt_malloc(&d_frames, cameras.size() * frame_floats * sizeof(float)));
      const float ms = single->RenderFrames(static_cast<float*>(d_frames), scene, cameras);
      std::cout << "Fly-through in batches: " << cameras.size() << " frames, " << ms / cameras.size() << "ms per frame ("
                << 1000.0 * cameras.size() / ms << " fps)" << std::endl;
      d_buffer.FreeGPU();
      batch_frames = d_frames;
      d_buffer.buffer = static_cast<float*>(d_frames) + (cameras.size() - 1) * frame_floats;
      renderTime = ms / cameras.size();
      times.push_back(renderTime);
    } else {
      for (size_t f = 0; f < cameras.size(); f++) {
        renderTime = render(d_buffer, scene, cameras[f]);
        times.push_back(renderTime);
      }
    }
    std::vector<float> sorted(times);
    std::sort(sorted.begin(), sorted.end());
    double sum = 0;
    for (float t : times) sum += t;
    std::cout << "Fly-through: " << times.size() << " frames, mean " << sum / times.size() << "ms, median "
              << sorted[sorted.size() / 2] << "ms, min " << sorted.front() << "ms, max " << sorted.back() << "ms ("
              << 1000.0 * times.size() / sum << " fps)" << std::endl;
  } else {
    for (int f = 0; f < frames; f++) renderTime = render(d_buffer, scene, camera);
  }
  std::cout << "Render completed in " << renderTime << "ms (" << 1000.0f / renderTime << " fps)" << std::endl;
  if (tiled) {
    std::cout << "Tile kernel times:";
    for (int g = 0; g < gpus; g++) std::cout << " " << tiled->TileKernelMs(g) << "ms";
    std::cout << std::endl;
  }
  std::cout << std::endl;
  if (!previewFile.empty()) {
    // what the interactive mode would put on screen (main.cu:175-176): Denoiser packs the colour
    // channels to RGBA8 point sprites; here they are unpacked into a PPM instead of drawn with GL
    Denoiser denoiser(width, height, threadsPerBlock);
    void* d_vertices = NULL;
    gpuErrchk(pt_malloc(&d_vertices, (size_t)width * height * 3 * sizeof(float)));
    denoiser.Denoise(d_buffer, static_cast<float*>(d_vertices));
    std::vector<float> vertices((size_t)width * height * 3);
    gpuErrchk(pt_memcpy_d2h(vertices.data(), d_vertices, vertices.size() * sizeof(float)));
    gpuErrchk(pt_free(d_vertices));
    FILE* f = fopen(previewFile.c_str(), "wb");
    if (!f) {
      std::cerr << "ERROR: cannot write " << previewFile << std::endl;
      return 1;
    }
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * height; i++) {
      unsigned char rgba[4];
      memcpy(rgba, &vertices[3 * i + 2], 4);
      fwrite(rgba, 1, 3, f);
    }
    fclose(f);
  }
  // save results (main.cu:186-192)
  OutputBuffer buffer(width, height);
  buffer.AllocateCPU();
  buffer.CopyFromGPU(d_buffer);
  buffer.SaveEXR(outputName + ".exr");
  if (!noBitmap) buffer.SaveBitmaps(outputName);
  buffer.FreeCPU();
  delete single;
  delete tiled;
  if (batch_frames) (void)pt_free(batch_frames);  // (d_buffer points into it)
  else d_buffer.FreeGPU();
  scene.Free();
  return 0;
}
