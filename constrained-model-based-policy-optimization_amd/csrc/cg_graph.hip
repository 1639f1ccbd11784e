// Conjugate-gradient solve of the CPO update as ONE call (utilities/trust_region.py:32-45 with the Fisher-vector
// product of policies/cpo_policy.py:168 as the operator).  Each iteration is four short kernels (direction pack,
// Fisher-vector product, ordered partial reduction, vector update); at N = 50 000 samples they run ~170 us and the
// launch gaps between them add ~15 %.  The iterations after the first are captured into a hipGraph (one graph launch
// replays them back to back); the graph is cached per policy handle and re-captured when any pointer or size changes.
// Single-GPU path only: with more ranks the Fisher-vector product is all-reduced between the two halves of an
// iteration by the host (torch.distributed), which a captured graph cannot contain.
#include "common.h"

#include <string.h>

#include <map>
#include <utility>

namespace {

struct CgKey {
  const void *h;
  cmbpo_pi_batch_t b;
  const float *vec;
  float *x, *r, *p;
  double *scal;
  double inv_n;
  float damping;
  int iters, P;
  hipStream_t s;
  const void *act;   // saved activations the captured Fisher-vector products read (or NULL)
  int path;          // arithmetic path of the captured kernels (cmbpo_set_pi_matrix_path)
};

struct CgGraph {
  CgKey key;
  hipGraphExec_t exec;
};

std::map<std::pair<const void *, const void *>, CgGraph> g_graphs;   // (policy handle, solution vector)
long g_graph_launches = 0, g_graph_captures = 0;
bool g_graph_ok = true;   // cleared when capture is not available; the eager loop is used from then on

int iteration(cmbpo_pi_t *h, const CgKey &k) {
  return cmbpo_pi_cg_iter(h, &k.b, k.inv_n, k.damping, k.x, k.r, k.p, k.scal, k.s);
}

}  // namespace

extern "C" int cmbpo_pi_cg_solve(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, const float *d_b, double inv_n, float damping,
                                 int iters, float *d_x, float *d_r, float *d_p, float *d_vec, double *d_scal,
                                 int use_graph, void *stream) {
  CMBPO_REQUIRE(h && b && d_b && d_x && d_r && d_p && d_vec && d_scal, "cmbpo_pi_cg_solve: NULL argument");
  CMBPO_REQUIRE(iters >= 1 && iters <= 1000, "cmbpo_pi_cg_solve: iters %d", iters);
  CgKey k;
  memset(&k, 0, sizeof(k));
  k.h = h; memcpy(&k.b, b, sizeof(k.b)); k.vec = d_vec; k.x = d_x; k.r = d_r; k.p = d_p; k.scal = d_scal;
  k.inv_n = inv_n; k.damping = damping; k.iters = iters; k.P = cmbpo_pi_num_params(h); k.s = (hipStream_t)stream;
  k.act = cmbpo_pi_act_token(h, b);
  k.path = cmbpo_get_pi_matrix_path();
  if (int rc = cmbpo_cg_init(k.P, d_b, d_x, d_r, d_p, d_scal, stream)) return rc;
  if (int rc = iteration(h, k)) return rc;            // eager: also performs any one-time kernel attribute set-up
  if (iters == 1) return cmbpo_pi_cg_commit(d_scal, stream);
  if (use_graph && g_graph_ok) {
    const std::pair<const void *, const void *> gk(h, d_x);
    auto it = g_graphs.find(gk);
    if (it != g_graphs.end() && memcmp(&it->second.key, &k, sizeof(k)) != 0) {
      (void)hipGraphExecDestroy(it->second.exec);
      g_graphs.erase(it);
      it = g_graphs.end();
    }
    if (it == g_graphs.end()) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured);
      // the instantiated graph is launched into the caller's stream
      static hipStream_t cap = nullptr;
      bool ok = cap != nullptr || hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess;
      int rc = CMBPO_OK;
      if (ok) ok = hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed) == hipSuccess;
      if (ok) {
        CgKey kc = k;
        kc.s = cap;
        for (int i = 1; i < iters && rc == CMBPO_OK; ++i) rc = iteration(h, kc);
        ok = hipStreamEndCapture(cap, &graph) == hipSuccess && rc == CMBPO_OK && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (!ok) {
        (void)hipGetLastError();
        if (g_graph_ok) fprintf(stderr, "cmbpo_pi_cg_solve: stream capture failed, the CG solves stay on the eager loop\n");
        g_graph_ok = false;               // capture not available here: stay on the eager loop
        if (rc != CMBPO_OK) return rc;
      } else {
        CgGraph cg;
        memcpy(&cg.key, &k, sizeof(k));   // byte copy: the cache check compares whole keys, padding bytes included
        cg.exec = exec;
        it = g_graphs.emplace(gk, cg).first;
        ++g_graph_captures;
      }
    }
    if (it != g_graphs.end()) {
      CMBPO_HIP_CHECK(hipGraphLaunch(it->second.exec, k.s));
      ++g_graph_launches;
      return cmbpo_pi_cg_commit(d_scal, stream);
    }
  }
  for (int i = 1; i < iters; ++i)
    if (int rc = iteration(h, k)) return rc;
  return cmbpo_pi_cg_commit(d_scal, stream);
}

// drop the cached graph of a handle (called before the handle is destroyed)
extern "C" void cmbpo_pi_cg_release(cmbpo_pi_t *h) {
  for (auto it = g_graphs.begin(); it != g_graphs.end();) {
    if (it->first.first == h) {
      (void)hipGraphExecDestroy(it->second.exec);
      it = g_graphs.erase(it);
    } else {
      ++it;
    }
  }
}

// graph launches so far, or -1 when stream capture turned out to be unavailable (diagnostics / tests)
extern "C" long cmbpo_pi_cg_graph_launches(void) { return g_graph_ok ? g_graph_launches : -1; }
// graphs captured so far: stays at one per (handle, solution vector) while the arguments do not change
extern "C" long cmbpo_pi_cg_graph_captures(void) { return g_graph_captures; }
