// run_real_all.cpp -- the reference's dataset driver (src/perception/src/app/run_real_all.cpp:70-273) above libhop, on the reference's
// own directory layout:
//   <base>/<model>/<record>/{rgbN.png, depthN.png, palm_in_baseN.txt, arm_left_link_7_t_N.txt}  ->  <record>/predict/<N>/model2scene.txt
//
//   run_real_all <config_autodataset.yaml> <assets_dir> <base_dir> [model_name]
//   RANK / WORLD_SIZE in the environment shard the frames (frame index mod world, as the Python runner does; the reference is one process);
//   HOP_FORCE=1 recomputes frames whose result exists (default: resume); HOP_INFLIGHT=N keeps N frames in flight on the device (N host
//   threads, each with its own estimator / hand / contexts; default 1 = the reference's sequential loop); LOCAL_RANK picks the device.
//   HOP_GATHER=1 (BASELINE configs[3], "frames sharded over the GPUs, RCCL gather of per-frame best pose"): at the end of its shard every
//   rank contributes its frames' poses to ONE hop_frames_allgather (ncclAllGather inside libhop.so); rank 0 writes them all to
//   <base>/<model>/model2scene_all.txt ("record index m00 ... m33" per frame).  The 128-byte RCCL id travels through the file
//   HOP_COMM_ID_FILE (default <base>/<model>/.hop_comm_id[.<MASTER_PORT>][.<HOP_RUN_ID>]: the ranks share the dataset directory already;
//   see comm_through_file for how a stale file and a failed rank are handled); one rank needs no communicator (HOP_GATHER_COMM=1
//   builds a 1-rank one all the same: the RCCL leg on a single-GPU box).  Same table as run_real_all.gather_frame_poses of the Python runner.
// assets_dir holds what the reference loads from PLY / OBJ / Boost archive / URDF files (download links): see hop::Assets (host/Frame.h).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <iostream>
#include <mutex>
#include <regex>
#include <thread>

#include <unistd.h>

#include "../Frame.h"

static std::vector<std::string> list_dir(const std::string& d, bool dirs) {
  std::vector<std::string> out;
  DIR* dp = opendir(d.c_str());
  if (!dp) return out;
  while (dirent* e = readdir(dp)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    struct stat st;
    if (stat((d + "/" + n).c_str(), &st) != 0) continue;
    if (dirs == (S_ISDIR(st.st_mode) != 0)) out.push_back(n);
  }
  closedir(dp);
  std::sort(out.begin(), out.end());
  return out;
}
static void mkdirs(const std::string& p) {
  std::string cur;
  std::istringstream ss(p);
  std::string part;
  if (!p.empty() && p[0] == '/') cur = "/";
  while (std::getline(ss, part, '/')) {
    if (part.empty()) continue;
    cur += part + "/";
    mkdir(cur.c_str(), 0755);
  }
}

struct FrameJob {
  std::string record, rec;
  int idx;
};

// The launcher's side of hop_comm_unique_id / hop_comm_create (include/hop.h: "the launcher hands the 128 bytes to the other ranks (any
// channel)"): rank 0 publishes the id in a file next to the frames (written under another name and renamed, so a reader never sees half
// of it), the others wait for it.  What keeps a file of ANOTHER run from being taken for this one's:
//   * the name carries the launch's MASTER_PORT and, if the launcher sets one, HOP_RUN_ID (a per-launch nonce);
//   * a waiting rank only accepts a file written after its own start minus HOP_COMM_STALE_S (default 120 s: the ranks of one launch start
//     together) -- the 128 bytes a killed run left behind are older than that and are ignored even if rank 0 has not yet removed them;
//   * rank 0 removes a left-over file when it starts and its own once the collective has returned (every rank has read it by then).
// A rank that fails in its shard publishes <id_file>.abort with the reason: ranks still waiting for the id stop with that reason instead of
// sitting out HOP_COMM_WAIT_S, and rank 0 does not publish an id any more (no rank is led into a collective that cannot complete).
static double file_age_before(const std::string& path, double t_ref) {  // seconds by which the file's mtime precedes t_ref (< 0: written later)
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return 1e300;
  return t_ref - ((double)st.st_mtim.tv_sec + 1e-9 * (double)st.st_mtim.tv_nsec);
}
static double wall_now() {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static const double g_start = wall_now();
static double stale_s() { return std::getenv("HOP_COMM_STALE_S") ? std::atof(std::getenv("HOP_COMM_STALE_S")) : 120.0; }
static bool abort_published(const std::string& id_file, std::string& why) {
  const std::string a = id_file + ".abort";
  // none, or one an earlier run left: a rank of THIS launch gives up after it has started, i.e. at most a launcher's start-up spread before
  // this rank's own start (10 s allowed) -- the id file's 120 s would let a re-launch right after a failure trip over the old marker
  if (file_age_before(a, g_start) > std::min(stale_s(), 10.0)) return false;
  std::ifstream f(a);
  std::getline(f, why);
  return true;
}
static void publish_abort(const std::string& id_file, int rank, const std::string& why) {
  std::ofstream f(id_file + ".abort.tmp" + std::to_string(rank));
  f << "rank " << rank << ": " << why << "\n";
  f.close();
  std::rename((id_file + ".abort.tmp" + std::to_string(rank)).c_str(), (id_file + ".abort").c_str());
}
static hop_comm* comm_through_file(const std::string& id_file, int device, int rank, int world) {
  unsigned char id[HOP_COMM_ID_BYTES];
  std::string why;
  if (rank == 0) {
    if (abort_published(id_file, why)) throw std::runtime_error("another rank gave up before the gather: " + why);
    if (hop_comm_unique_id(id) != HOP_OK) throw std::runtime_error(std::string("hop_comm_unique_id: ") + hop_comm_last_error(nullptr));
    {
      std::ofstream f(id_file + ".tmp", std::ios::binary);
      f.write(reinterpret_cast<const char*>(id), sizeof id);
      if (!f) throw std::runtime_error("cannot write " + id_file);
    }
    if (std::rename((id_file + ".tmp").c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish " + id_file);
  } else {
    const double wait_s = std::getenv("HOP_COMM_WAIT_S") ? std::atof(std::getenv("HOP_COMM_WAIT_S")) : 3600.0;  // rank 0 may still be in its shard
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      if (abort_published(id_file, why)) throw std::runtime_error("another rank gave up before the gather: " + why);
      if (file_age_before(id_file, g_start) <= stale_s()) {  // (an older file is a killed run's: not this launch's id)
        std::ifstream f(id_file, std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(id), sizeof id)) break;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s)
        throw std::runtime_error("rank 0 did not publish the RCCL id in " + id_file);
      usleep(20000);
    }
  }
  hop_comm* comm = nullptr;
  if (hop_comm_create(device, id, rank, world, &comm) != HOP_OK) throw std::runtime_error(std::string("hop_comm_create: ") + hop_comm_last_error(nullptr));
  return comm;
}

static std::string g_id_file;  // set once the gather of a multi-rank run is known to be wanted: where a failure is announced
static int g_rank = 0;

int main(int argc, char** argv) {
  if (argc < 4) {
    std::cout << "usage: run_real_all <config.yaml> <assets_dir> <base_dir> [model_name]\n";
    return 2;
  }
  try {
    ConfigParser cfg0(argv[1]);
    const hop::Assets assets(argv[2]);
    const std::string base = argv[3], model_name = argc > 4 ? argv[4] : cfg0.model_name;
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0, world = std::getenv("WORLD_SIZE") ? std::max(1, std::atoi(std::getenv("WORLD_SIZE"))) : 1;
    const bool force = std::getenv("HOP_FORCE") != nullptr;
    const int inflight = std::getenv("HOP_INFLIGHT") ? std::max(1, std::atoi(std::getenv("HOP_INFLIGHT"))) : 1;
    const int device = std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : 0;
    const std::string mdir = base + "/" + model_name;
    // this rank's frames that have no result yet (the reference walks the same directories in the same order, :70-115)
    std::vector<FrameJob> jobs, all_frames, mine;  // all_frames: every rank lists the same directory, so its order numbers the frames for all
    int n_skipped = 0;
    const bool gather = std::getenv("HOP_GATHER") != nullptr;
    std::string id_file = std::getenv("HOP_COMM_ID_FILE") ? std::getenv("HOP_COMM_ID_FILE") : mdir + "/.hop_comm_id";
    if (!std::getenv("HOP_COMM_ID_FILE") && std::getenv("MASTER_PORT")) id_file += std::string(".") + std::getenv("MASTER_PORT");
    if (!std::getenv("HOP_COMM_ID_FILE") && std::getenv("HOP_RUN_ID")) id_file += std::string(".") + std::getenv("HOP_RUN_ID");
    // HOP_GATHER_COMM=1: go through the communicator even with ONE rank (the ncclAllGather leg then runs on a 1-rank RCCL communicator:
    // what a single-GPU box can exercise of configs[3]'s collective)
    const bool use_comm = gather && (world > 1 || std::getenv("HOP_GATHER_COMM") != nullptr);
    g_id_file = use_comm ? id_file : std::string();
    g_rank = rank;
    if (use_comm && rank == 0) {
      std::remove(id_file.c_str());
      // the abort marker only if it is an EARLIER run's (the age rule abort_published reads it by): a peer of this launch that failed at once
      // may have published its reason before rank 0 got here, and removing it would leave every rank waiting for a collective that cannot complete
      if (file_age_before(id_file + ".abort", g_start) > std::min(stale_s(), 10.0)) std::remove((id_file + ".abort").c_str());
    }
    const std::regex rgb_re("rgb([0-9]+)\\..*");
    for (const std::string& record : list_dir(mdir, true)) {
      const std::string rec = mdir + "/" + record;
      std::vector<int> frames;  // run_real_all.cpp:76-97: the index sits between "rgb" and the first "."
      for (const std::string& f : list_dir(rec, false)) {
        std::smatch m;
        if (std::regex_match(f, m, rgb_re)) frames.push_back(std::atoi(m[1].str().c_str()));
      }
      std::sort(frames.begin(), frames.end());
      for (int idx : frames) {
        all_frames.push_back({record, rec, idx});
        if (idx % world != rank) continue;
        mine.push_back({record, rec, idx});
        std::ifstream ex(rec + "/predict/" + std::to_string(idx) + "/model2scene.txt");
        if (ex && !force) ++n_skipped;
        else jobs.push_back({record, rec, idx});
      }
    }
    // Frames are independent (the reference resets its estimator and hand between them, :265-266), so HOP_INFLIGHT workers -- each with
    // its own estimator, hand and pair of contexts (= HIP streams) on this rank's device, all built ONCE as the reference builds its
    // pair (:56-68) -- take them from one counter: the host work of a frame (PNG decoding, base selection, the PSO bookkeeping) then
    // runs beside the kernels of the other frames.  Same results as one worker: nothing is shared but the read-only assets.
    std::atomic<size_t> next{0};
    std::atomic<int> n_done{0};
    std::mutex io;
    std::string first_error;
    double ms_total = 0;
    std::vector<std::pair<std::string, double>> stage_total;  // HOP_APP_TIMING=1: host wall time per stage, summed over the frames
    const auto wall0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
      try {
        ConfigParser cfg(argv[1]);
        const hop::Calibration cal(cfg);
        PoseEstimator est(&cfg, assets.model, assets.model001, device);
        HandT42 hand(&cfg, est.ctx());
        assets.addTo(hand);
        hop_ctx* icp_ctx = nullptr;
        hop::check(hop_ctx_create(device, &icp_ctx), nullptr, "hop_ctx_create");
        hand.setHandbaseIcpContext(icp_ctx);
        for (size_t k = next.fetch_add(1); k < jobs.size(); k = next.fetch_add(1)) {
          const FrameJob& j = jobs[k];
          const std::string out_dir = j.rec + "/predict/" + std::to_string(j.idx), out = out_dir + "/model2scene.txt";
          const auto t0 = std::chrono::steady_clock::now();
          const Mat4 leftarm_in_base = hop::parse_pose_txt(j.rec + "/arm_left_link_7_t_" + std::to_string(j.idx) + ".txt");
          const Mat4 palm_in_baselink = hop::parse_pose_txt(j.rec + "/palm_in_base" + std::to_string(j.idx) + ".txt");
          std::vector<uint16_t> depth;
          int H = 0, W = 0;
          hop::read_png16(j.rec + "/depth" + std::to_string(j.idx) + ".png", depth, H, W);
          hop::FrameInfo info;
          const Mat4 pose = hop::process_frame(cfg, assets, est, hand, depth, H, W, cal.K9, cal.handbaseInCam(leftarm_in_base, palm_in_baselink), 0.001, true, true, &info);
          mkdirs(out_dir);
          {
            std::ofstream ff(out + ".tmp");  // a killed run never leaves a half-written result
            ff.precision(9);
            for (int r = 0; r < 4; ++r)
              for (int c = 0; c < 4; ++c) ff << pose.m[4 * r + c] << (c == 3 ? "\n" : " ");
          }
          std::rename((out + ".tmp").c_str(), out.c_str());
          const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          std::lock_guard<std::mutex> lk(io);
          ms_total += ms;
          ++n_done;
          for (size_t q = 0; q < info.stage_ms.size(); ++q) {
            if (q >= stage_total.size() || stage_total[q].first != info.stage_ms[q].first) stage_total.insert(stage_total.begin() + q, {info.stage_ms[q].first, 0.0});
            stage_total[q].second += info.stage_ms[q].second;
          }
          std::printf("%s/%d: %.1f ms, %d hand-region points, %d object points, %d hypotheses after ICP, score %.2f\n", j.record.c_str(), j.idx, ms,
                      info.n_hand_region, info.n_object_segment, info.n_after_icp, info.score);
        }
        hop_ctx_destroy(icp_ctx);
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(io);
        if (first_error.empty()) first_error = e.what();
        next.store(jobs.size());  // the other workers finish their frame and stop
      }
    };
    std::vector<std::thread> pool;
    if (!jobs.empty()) {  // (nothing left to do: no estimator, no context, no device is touched)
      for (int w = 1; w < std::min<int>(inflight, (int)jobs.size()); ++w) pool.emplace_back(worker);
      worker();
    }
    for (std::thread& t : pool) t.join();
    if (!first_error.empty()) throw std::runtime_error(first_error);
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    const int nd = n_done.load();
    std::printf("rank %d of %d: %d frames written (%.1f ms per frame alone, %d in flight: %.1f ms of wall per frame), %d resumed\n", rank, world, nd,
                nd ? ms_total / nd : 0.0, inflight, nd ? wall / nd : 0.0, n_skipped);
    if (std::getenv("HOP_APP_TIMING") && nd)
      for (const auto& st : stage_total) std::printf("  %-45s %8.2f ms per frame\n", st.first.c_str(), st.second / nd);
    if (gather) {
      // rows { frame number, pose[16] } of this rank's frames (computed now or by an earlier, resumed run: the result files are the
      // record of both), padded by the library to the largest shard, which every rank derives from the same list and the same rule
      std::vector<int> counts(world, 0);
      for (const FrameJob& j : all_frames) ++counts[j.idx % world];
      const int rows_per_rank = std::max(1, *std::max_element(counts.begin(), counts.end()));
      std::vector<float> rows(mine.size() * HOP_FRAME_ROW_FLOATS);
      size_t r = 0, number = 0;
      for (const FrameJob& j : all_frames) {
        if (j.idx % world == rank) {
          const Mat4 T = hop::parse_pose_txt(j.rec + "/predict/" + std::to_string(j.idx) + "/model2scene.txt");
          rows[r * HOP_FRAME_ROW_FLOATS] = (float)number;
          std::copy(T.m, T.m + 16, rows.begin() + r * HOP_FRAME_ROW_FLOATS + 1);
          ++r;
        }
        ++number;
      }
      std::vector<float> table;
      if (!use_comm) table = rows;  // one rank: its rows are the table
      else {
        hop_comm* comm = comm_through_file(id_file, device, rank, world);
        table.assign((size_t)world * rows_per_rank * HOP_FRAME_ROW_FLOATS, 0.f);
        const int rc = hop_frames_allgather(comm, rows.data(), (int)mine.size(), rows_per_rank, table.data());
        const std::string why = rc == HOP_OK ? "" : std::string(hop_strerror(rc)) + " " + hop_comm_last_error(comm);
        hop_comm_destroy(comm);
        if (rank == 0) std::remove(id_file.c_str());
        if (rc != HOP_OK) throw std::runtime_error("hop_frames_allgather: " + why);
      }
      std::vector<const float*> by_number(all_frames.size(), nullptr);
      for (size_t q = 0; q < table.size() / HOP_FRAME_ROW_FLOATS; ++q) {
        const float* row = table.data() + q * HOP_FRAME_ROW_FLOATS;
        if (row[0] >= 0 && (size_t)row[0] < by_number.size()) by_number[(size_t)row[0]] = row;
      }
      size_t n_have = 0;
      for (const float* row : by_number) n_have += row != nullptr;
      if (n_have != all_frames.size()) throw std::runtime_error("the gathered table holds " + std::to_string(n_have) + " of " + std::to_string(all_frames.size()) + " frames");
      if (rank == 0) {
        const std::string out = mdir + "/model2scene_all.txt";
        {
          std::ofstream ff(out + ".tmp");
          ff.precision(9);
          for (size_t q = 0; q < all_frames.size(); ++q) {
            ff << all_frames[q].record << " " << all_frames[q].idx;
            for (int e = 0; e < 16; ++e) ff << " " << by_number[q][1 + e];
            ff << "\n";
          }
        }
        std::rename((out + ".tmp").c_str(), out.c_str());
      }
      std::printf("rank %d of %d: poses of %zu frames gathered (%zu from this rank, %d rows per rank%s)\n", rank, world, all_frames.size(), mine.size(), rows_per_rank,
                  use_comm ? ", through hop_frames_allgather" : "");
    }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    if (!g_id_file.empty()) publish_abort(g_id_file, g_rank, e.what());  // the ranks waiting for the gather stop too
    return 3;
  }
}
