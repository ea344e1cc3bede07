// placeholder until the joint-fit kernels land
#include "lc_common.h"
struct lc_joint { lc_ctx *ctx; };
extern "C" {
int lc_joint_supported(int, int) { return 0; }
int lc_joint_create(lc_ctx *ctx, int, int, int, int, const float *, const float *, const float *, lc_joint **) { if (ctx) ctx->err = "joint fit not built"; return LC_ERR_UNSUPPORTED; }
void lc_joint_destroy(lc_joint *) {}
int lc_joint_set_param(lc_joint *, int, const float *, int) { return LC_ERR_UNSUPPORTED; }
int lc_joint_get_param(lc_joint *, int, float *, int) { return LC_ERR_UNSUPPORTED; }
int lc_joint_set_free(lc_joint *, const int32_t *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_set_loss(lc_joint *, const lc_joint_loss_cfg *, const float *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_propagate_noise(lc_joint *, float *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_loss_grad(lc_joint *, float *, float *const *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_model(lc_joint *, float *, float *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_deconvolved(lc_joint *, int, float *, float *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_run_adabelief(lc_joint *, int, const lc_adabelief_cfg *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_get_loss_history(lc_joint *, float *, int) { return LC_ERR_UNSUPPORTED; }
int lc_joint_iterations_done(lc_joint *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_fisher_flux_sigma(lc_joint *, float *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_step_local(lc_joint *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_shared_buffer_dev(lc_joint *, void **, int *) { return LC_ERR_UNSUPPORTED; }
int lc_joint_step_update(lc_joint *, const lc_adabelief_cfg *) { return LC_ERR_UNSUPPORTED; }
}
