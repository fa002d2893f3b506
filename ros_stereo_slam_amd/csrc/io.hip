// io.hip -- the data formats either side of the hot path: KITTI pose
// files, the reference's trajectory.csv, trajectory error metrics, the map / pose message
// conversions of rosPublish and the PLY dump.  Host code only; nothing here touches the GPU.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "svo_internal.h"

namespace {

void mat3_mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// relative motion  A^-1 B  of two camera-to-world poses
void rel_pose(const double *Ra, const double *ta, const double *Rb, const double *tb, double *R, double *t)
{
    double Rat[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Rat[3 * i + j] = Ra[3 * j + i];
    mat3_mul(Rat, Rb, R);
    const double d[3] = {tb[0] - ta[0], tb[1] - ta[1], tb[2] - ta[2]};
    for (int i = 0; i < 3; i++)
        t[i] = Rat[3 * i] * d[0] + Rat[3 * i + 1] * d[1] + Rat[3 * i + 2] * d[2];
}
// Rodrigues vector of a rotation matrix (cv::Rodrigues(R) -> rvec)
void rot_to_rvec(const double *R, double *r)
{
    const double c = ((R[0] + R[4] + R[8]) - 1.) * 0.5;
    const double cc = c > 1 ? 1 : (c < -1 ? -1 : c);
    const double theta = acos(cc);
    const double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    if (s < 1e-5) {
        if (cc > 0) {
            r[0] = r[1] = r[2] = 0;
            return;
        }
        // theta ~ pi: axis from the diagonal
        double ax = sqrt(fmax((R[0] + 1) * 0.5, 0.)), ay = sqrt(fmax((R[4] + 1) * 0.5, 0.)) * (R[1] < 0 ? -1 : 1),
               az = sqrt(fmax((R[8] + 1) * 0.5, 0.)) * (R[2] < 0 ? -1 : 1);
        if (fabs(ax) < fabs(ay) && fabs(ax) < fabs(az) && ((R[5] > 0) != (ay * az > 0)))
            az = -az;
        const double n = sqrt(ax * ax + ay * ay + az * az);
        const double k = n > 0 ? theta / n : 0;
        r[0] = ax * k;
        r[1] = ay * k;
        r[2] = az * k;
        return;
    }
    const double k = theta / (2 * s);
    r[0] = rx * k;
    r[1] = ry * k;
    r[2] = rz * k;
}

}  // namespace

extern "C" {

int svo_io_read_kitti_poses(const char *path, double *Rt12, int capacity, int *n)
{
    SVO_CHECK_ARG(path && n && capacity >= 0 && (Rt12 || capacity == 0));
    *n = 0;
    FILE *f = fopen(path, "r");
    if (!f) {
        svo_set_error("cannot open %s", path);
        return SVO_ERR_ARG;
    }
    int count = 0;
    double v[12];
    for (;;) {
        int got = 0;
        for (; got < 12; got++)
            if (fscanf(f, "%lf", &v[got]) != 1)
                break;
        if (got == 0)
            break;
        if (got != 12) {
            fclose(f);
            svo_set_error("%s: pose %d has %d of 12 numbers", path, count, got);
            return SVO_ERR_ARG;
        }
        if (count < capacity)
            memcpy(Rt12 + (size_t)count * 12, v, sizeof(v));
        count++;
    }
    fclose(f);
    *n = count;  // the number in the file, even when it exceeds the capacity
    return SVO_OK;
}

int svo_io_write_kitti_poses(const char *path, const double *R9s, const double *t3s, int n)
{
    SVO_CHECK_ARG(path && n >= 0 && (n == 0 || (R9s && t3s)));
    FILE *f = fopen(path, "w");
    if (!f) {
        svo_set_error("cannot open %s", path);
        return SVO_ERR_ARG;
    }
    for (int i = 0; i < n; i++) {
        const double *R = R9s + (size_t)i * 9, *t = t3s + (size_t)i * 3;
        fprintf(f, "%.9e %.9e %.9e %.9e %.9e %.9e %.9e %.9e %.9e %.9e %.9e %.9e\n", R[0], R[1], R[2], t[0], R[3], R[4],
                R[5], t[1], R[6], R[7], R[8], t[2]);
    }
    fclose(f);
    return SVO_OK;
}

int svo_io_trajectory_csv(const char *path, const float *rows8, int n_rows, int create)
{
    SVO_CHECK_ARG(path && n_rows >= 0 && (n_rows == 0 || rows8));
    FILE *f = fopen(path, create ? "w" : "a");
    if (!f) {
        svo_set_error("cannot open %s", path);
        return SVO_ERR_ARG;
    }
    if (create)
        fprintf(f, "Idx,Xm,Ym,Zm,Xgt,Ygt,Zgt,Const\n");
    for (int i = 0; i < n_rows; i++) {
        for (int j = 0; j < 8; j++)
            fprintf(f, "%g,", (double)rows8[(size_t)i * 8 + j]);  // operator<<(float): %g, 6 significant digits
        fprintf(f, "\n");
    }
    fclose(f);
    return SVO_OK;
}

int svo_eval_ate_rmse(const double *t3_est, const double *t3_gt, int n, double *rmse)
{
    SVO_CHECK_ARG(t3_est && t3_gt && n > 0 && rmse);
    double s = 0;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const double d = t3_est[3 * i + k] - t3_gt[3 * i + k];
            s += d * d;
        }
    *rmse = sqrt(s / n);
    return SVO_OK;
}

int svo_eval_rpe(const double *R9_est, const double *t3_est, const double *R9_gt, const double *t3_gt, int n,
                 int delta, double *trans_rmse, double *rot_rmse)
{
    SVO_CHECK_ARG(R9_est && t3_est && R9_gt && t3_gt && delta >= 1 && n > delta);
    double st = 0, sr = 0;
    int cnt = 0;
    for (int i = 0; i + delta < n; i++) {
        const int j = i + delta;
        double Rp[9], tp[3], Rq[9], tq[3], Re[9], te[3];
        rel_pose(R9_est + 9 * i, t3_est + 3 * i, R9_est + 9 * j, t3_est + 3 * j, Rp, tp);
        rel_pose(R9_gt + 9 * i, t3_gt + 3 * i, R9_gt + 9 * j, t3_gt + 3 * j, Rq, tq);
        rel_pose(Rq, tq, Rp, tp, Re, te);  // Q^-1 P
        st += te[0] * te[0] + te[1] * te[1] + te[2] * te[2];
        double c = ((Re[0] + Re[4] + Re[8]) - 1.) * 0.5;
        c = c > 1 ? 1 : (c < -1 ? -1 : c);
        const double a = acos(c);
        sr += a * a;
        cnt++;
    }
    if (trans_rmse)
        *trans_rmse = sqrt(st / cnt);
    if (rot_rmse)
        *rot_rmse = sqrt(sr / cnt);
    return SVO_OK;
}

int svo_ros_map_points(const float *xyz, const float *bgr, int n, float *xyz_out, uint8_t *rgb_out)
{
    if (n < 0 || (n > 0 && (!xyz || !xyz_out)) || (rgb_out && !bgr)) {
        svo_set_error("svo_ros_map_points: bad arguments");
        return SVO_ERR_ARG;
    }
    int k = 0;
    const float mul = 0.1f;  // src/rosFuncs.cpp:46
    for (int i = 0; i < n; i++) {
        if (-1 * xyz[3 * i + 2] > 500)  // :54
            continue;
        xyz_out[3 * k] = xyz[3 * i] * mul;
        xyz_out[3 * k + 1] = xyz[3 * i + 2] * mul;
        xyz_out[3 * k + 2] = -1 * xyz[3 * i + 1] * mul;
        if (rgb_out) {  // clPt.r = colorMap.z, .g = .y, .b = .x (:60), float -> uint8 as PCL's members
            rgb_out[3 * k] = (uint8_t)bgr[3 * i + 2];
            rgb_out[3 * k + 1] = (uint8_t)bgr[3 * i + 1];
            rgb_out[3 * k + 2] = (uint8_t)bgr[3 * i];
        }
        k++;
    }
    return k;
}

int svo_ros_pose(const double *R9, const double *t3, double *pos3, double *quat4)
{
    SVO_CHECK_ARG(R9 && t3 && pos3 && quat4);
    pos3[0] = t3[0] * 0.1;   // src/rosFuncs.cpp:71-75
    pos3[1] = t3[2] * 0.1;
    pos3[2] = t3[1] * 0.1 * -1;
    double r[3];
    rot_to_rvec(R9, r);  // include/monoUtils.h:216
    // quat = AngleAxis(r0, X) * AngleAxis(r1, Y) * AngleAxis(r2, Z)  (:221-224)
    const double cx = cos(r[0] * 0.5), sx = sin(r[0] * 0.5), cy = cos(r[1] * 0.5), sy = sin(r[1] * 0.5),
                 cz = cos(r[2] * 0.5), sz = sin(r[2] * 0.5);
    // qx * qy
    const double aw = cx * cy, ax = sx * cy, ay = cx * sy, az = sx * sy;
    // (qx qy) * qz, qz = (0, 0, sz, cz)
    const double qw = aw * cz - az * sz, qx = ax * cz + ay * sz, qy = ay * cz - ax * sz, qz = aw * sz + az * cz;
    quat4[0] = qx;        // src/rosFuncs.cpp:88-91
    quat4[1] = qz;
    quat4[2] = qy * -1;
    quat4[3] = qw;
    return SVO_OK;
}

int svo_io_write_ply(const char *path, const float *xyz, const uint8_t *rgb, int n)
{
    SVO_CHECK_ARG(path && n >= 0 && (n == 0 || xyz));
    FILE *f = fopen(path, "wb");
    if (!f) {
        svo_set_error("cannot open %s", path);
        return SVO_ERR_ARG;
    }
    fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment ros_stereo_slam_amd map\nelement vertex %d\n"
               "property float x\nproperty float y\nproperty float z\n", n);
    if (rgb)
        fprintf(f, "property uchar red\nproperty uchar green\nproperty uchar blue\n");
    fprintf(f, "end_header\n");
    for (int i = 0; i < n; i++) {
        fwrite(xyz + (size_t)3 * i, sizeof(float), 3, f);
        if (rgb)
            fwrite(rgb + (size_t)3 * i, 1, 3, f);
    }
    fclose(f);
    return SVO_OK;
}

}  // extern "C"
