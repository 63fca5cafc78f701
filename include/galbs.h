/*
 * galbs.h — C ABI of the MI355X (gfx950) SMPL linear-blend-skinning kernels.
 *
 * Replaces, on the render-and-fit hot path of the reference:
 *
 *   joint transforms   /root/reference/submodules/smplx/lbs.py:216 (batch_rodrigues, :299-333),
 *                      :232 (batch_rigid_transform, :349-405) as reached from
 *                      SMPL.forward /root/reference/submodules/smplx/body_models.py:369-383
 *                      (A[:, :, :3, 3] += transl) and SMPLX.forward :1234-1291,
 *                      followed by `cano2live = matmul(A, inv_mats)`
 *                      /root/reference/model/avatar_model.py:296
 *   point skinning     the two einsums /root/reference/model/avatar_model.py:311-314
 *                      (pt_mats = sum_j w_nj M_j ; x' = R_n (x + res) + t_n)
 *
 * The vertex path of lbs() (pose blend shapes lbs.py:221, vertex skinning :239-247,
 * vertex_joint_selector body_models.py:375) is dead work for this path and is not
 * reproduced: the rest-pose joints J(betas) are an input (constant during training).
 *
 * Conventions: device pointers, caller-owned buffers, everything enqueued on `stream`
 * (hipStream_t as void*), row-major 4x4 matrices exactly as the torch tensors the
 * reference holds ([...,4,4] contiguous), return 0 = OK, text via galbs_last_error().
 */
#ifndef GALBS_H
#define GALBS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GALBS_ABI_VERSION 1
#define GALBS_MAX_JOINTS 64     /* SMPL = 24, SMPL-X = 55 */

/*
 * Forward joint transforms for B frames.
 *   pose        [B, J*3]  axis-angle, joint 0 = global orientation
 *   transl      [B, 3] or NULL
 *   joints_rest [J, 3]    rest-pose joint locations J(betas)
 *   parents     [J] int32 kinematic tree, parents[0] = -1, parents[i] < i
 *   inv_mats    [J,4,4] (inv_batch_stride = 0) or [B,J,4,4] (inv_batch_stride = J*16):
 *               inverse of the canonical-pose A (avatar_model.py:89)
 * Outputs
 *   A           [B,J,4,4]  SMPLOutput.A including transl (body_models.py:383)
 *   cano2live   [B,J,4,4]  A @ inv_mats                  (avatar_model.py:296)
 *   saved       [B, galbs_joint_saved_floats(J)] floats kept for backward
 */
size_t galbs_joint_saved_floats(int32_t J);

int galbs_joint_transforms_fwd(int32_t B, int32_t J,
                               const float* pose, const float* transl,
                               const float* joints_rest, const int32_t* parents,
                               const float* inv_mats, int64_t inv_batch_stride,
                               float* A, float* cano2live, float* saved, void* stream);

/*
 * Backward of the above w.r.t. pose and transl.
 *   dL_dcano2live [B,J,4,4] or NULL, dL_dA [B,J,4,4] or NULL (at least one non-NULL)
 *   dL_dpose [B,J*3], dL_dtransl [B,3] (NULL allowed) — fully overwritten.
 */
int galbs_joint_transforms_bwd(int32_t B, int32_t J,
                               const float* pose, const float* joints_rest,
                               const int32_t* parents,
                               const float* inv_mats, int64_t inv_batch_stride,
                               const float* saved,
                               const float* dL_dcano2live, const float* dL_dA,
                               float* dL_dpose, float* dL_dtransl, void* stream);

/*
 * Fused point skinning for B frames of N points.
 *   points   [N,3] (pts_batch_stride = 0) or [B,N,3] (pts_batch_stride = N*3)
 *   res      same, with res_batch_stride; may be NULL (= zeros)
 *   weights  [N,J] (w_batch_stride = 0) or [B,N,J] (w_batch_stride = N*J)
 *   mats     [B,J,4,4] cano2live
 *   out      [B,N,3]   out[b,n] = T[:3,:3] (points+res) + T[:3,3],  T = sum_j w[n,j] mats[b,j]
 */
int galbs_skin_fwd(int32_t B, int32_t N, int32_t J,
                   const float* points, int64_t pts_batch_stride,
                   const float* res, int64_t res_batch_stride,
                   const float* weights, int64_t w_batch_stride,
                   const float* mats, float* out, void* stream);

/*
 * Backward: dL_dout [B,N,3] ->
 *   dL_dres  [B,N,3] (also the gradient w.r.t. points), NULL allowed
 *   dL_dmats [B,J,4,4], NULL allowed; fully overwritten (bottom rows zero)
 */
int galbs_skin_bwd(int32_t B, int32_t N, int32_t J,
                   const float* points, int64_t pts_batch_stride,
                   const float* res, int64_t res_batch_stride,
                   const float* weights, int64_t w_batch_stride,
                   const float* mats, const float* dL_dout,
                   float* dL_dres, float* dL_dmats, void* stream);

const char* galbs_last_error(void);
int galbs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GALBS_H */
