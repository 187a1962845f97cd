"""Evaluation half of lib/dataset/LM6D_REFINE.py — the tables the reference's test loop prints after the refinement
iterations (deepim/core/tester.py:560-580 → LM6D_REFINE.evaluate_pose :278-370, evaluate_pose_add :372-512,
evaluate_pose_arp_2d :514-674): per class and per refinement iteration the (n°, n cm) accuracies for n = 1…10, the
ADD / ADD-S accuracies at 0.02 / 0.05 / 0.10 of the object diameter with the area under the accuracy-threshold curve,
and the 2 / 5 / 10 / 20 px reprojection accuracies with their curve.

Same method names, argument order, `all_poses_est[cls][iter]` / `all_poses_gt[cls][0]` containers and — line for line —
the same `logger.info` output as the reference; in addition every method RETURNS its tables.  The per-pose metrics (re,
te, ADD, ADI, arp_2d of lib/utils/pose_error.py) are one `deepim_pose_error` launch per class on the device; the
accumulation is host-side numpy.  The dataset half of the reference class (image / pose file indexing) is out of scope.

Differences: no matplotlib figures (the curves are returned and pickled as the reference pickles them); Simpson's rule
comes from `scipy.integrate.simpson` (`simps` was removed from SciPy 1.14).
"""
import logging
import os
import pickle

import numpy as np

_LOG = logging.getLogger("mx_deepim_amd")
SYM_ADI = ("eggbox", "glue", "bowl", "cup")                   # LM6D_REFINE.py:425
RT_Z = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0]], np.float32)   # :307, :565: half a turn about the object's z


def _simpson(y, dx):
    try:
        from scipy.integrate import simpson
    except ImportError:                                        # SciPy < 1.6
        from scipy.integrate import simps as simpson
    return simpson(y, dx=dx)


def _turn_z(poses):
    """se3_mul(pose, RT_z) (lib/utils/projection.py:23-43) for a stack of (3,4) poses: R·diag(-1,-1,1), t unchanged."""
    out = np.array(poses, np.float32, copy=True)
    out[:, :, 0] *= -1
    out[:, :, 1] *= -1
    return out


class LM6D_REFINE(object):
    def __init__(self, classes, points, diameters, ctx=None, logger=None):
        """classes: list of names; points: name → (N,3) model points; diameters: name → metres.
        ctx: runtime.Context for the device metrics (default: device 0, created on first use)."""
        self.classes = list(classes)
        self.num_classes = len(self.classes)
        self._points = points
        self._diameters = diameters
        self._ctx = ctx
        self.logger = logger or _LOG

    # ------------------------------------------------------------------------------------------ per-pose metrics
    def _device_metrics(self, cls_name, est, gt, K):
        """est, gt (n,3,4) → (n,5) [re°, te, add, adi, arp_2d] from ONE deepim_pose_error launch."""
        from ...runtime import Context, lib
        if self._ctx is None:
            self._ctx = Context.default()
        ctx = self._ctx
        n = len(est)
        pts = np.ascontiguousarray(np.asarray(self._points[cls_name], np.float32).T)          # (3,N)
        out = ctx.empty((n, 5))
        lib.deepim_pose_error(ctx.handle, out, ctx.array(np.ascontiguousarray(est, np.float32)),
                              ctx.array(np.ascontiguousarray(gt, np.float32)), ctx.array(pts), 1,
                              np.ascontiguousarray(K, np.float32), n, pts.shape[1])
        return out.asnumpy().astype(np.float64)

    def pose_metrics(self, config, all_poses_est, all_poses_gt):
        """name → (num_iter, n, 4) float64: [rot° , trans m] as evaluate_pose uses them (eggbox half-turn rule on
        calc_rt_dist_m > 90°, :304-309), the ADD / ADD-S error as evaluate_pose_add uses it (:423-428: ADI for
        eggbox/glue/bowl/cup, no half-turn rule), arp_2d as evaluate_pose_arp_2d uses it (rule on re > 90°, :563-571)."""
        num_iter = config.TEST.test_iter
        K = config.dataset.INTRINSIC_MATRIX
        res = {}
        for ci, name in enumerate(self.classes):
            if not (len(all_poses_est[ci][0]) and len(all_poses_gt[ci][0])):
                continue
            gt = np.asarray(all_poses_gt[ci][0], np.float32).reshape(-1, 3, 4)
            n = len(gt)
            est = np.asarray([all_poses_est[ci][it] for it in range(num_iter)], np.float32).reshape(num_iter * n, 3, 4)
            gtt = np.tile(gt, (num_iter, 1, 1))
            m = self._device_metrics(name, est, gtt, K)
            out = np.stack([m[:, 0], m[:, 1], m[:, 3] if name in SYM_ADI else m[:, 2], m[:, 4]], 1)
            if name == "eggbox":
                flip = m[:, 0] > 90
                if flip.any():
                    ms = self._device_metrics(name, _turn_z(est[flip]), gtt[flip], K)
                    out[flip, 0], out[flip, 1], out[flip, 3] = ms[:, 0], ms[:, 1], ms[:, 4]
            res[name] = out.reshape(num_iter, n, 4)
        return res

    # ------------------------------------------------------------------------------- (n°, n cm), :278-370
    def evaluate_pose(self, config, all_poses_est, all_poses_gt, metrics=None):
        log = self.logger
        log.info("evaluating pose")
        metrics = metrics if metrics is not None else self.pose_metrics(config, all_poses_est, all_poses_gt)
        rot_thresh_list = np.arange(1, 11, 1)
        trans_thresh_list = np.arange(0.01, 0.11, 0.01)
        num_metric = len(rot_thresh_list)
        num_iter = config.TEST.test_iter
        rot_acc = np.zeros((self.num_classes, num_iter, num_metric))
        trans_acc = np.zeros((self.num_classes, num_iter, num_metric))
        space_acc = np.zeros((self.num_classes, num_iter, num_metric))
        num_valid_class = 0
        row = "{:>16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}"
        head = "{:>24}: {:>7}, {:>7}, {:>7}".format("[rot_thresh, trans_thresh", "RotAcc", "TraAcc", "SpcAcc")
        show_list = [1, 4, 9]
        for cls_idx, cls_name in enumerate(self.classes):
            if cls_name not in metrics:
                continue
            num_valid_class += 1
            m = metrics[cls_name]
            for iter_i in range(num_iter):
                r, t = m[iter_i, :, 0:1], m[iter_i, :, 1:2]
                rot_acc[cls_idx, iter_i] = np.mean(r < rot_thresh_list[None, :], axis=0)
                trans_acc[cls_idx, iter_i] = np.mean(t < trans_thresh_list[None, :], axis=0)
                space_acc[cls_idx, iter_i] = np.mean(np.logical_and(r < rot_thresh_list[None, :], t < trans_thresh_list[None, :]),
                                                     axis=0)
            log.info("------------ {} -----------".format(cls_name))
            log.info(head)
            for iter_i in range(num_iter):
                log.info("** iter {} **".format(iter_i + 1))
                log.info("{:<16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}".format(
                    "average_accuracy", "[{:>2}, {:>5.2f}]".format(-1, -1), np.mean(rot_acc[cls_idx, iter_i, :]) * 100,
                    np.mean(trans_acc[cls_idx, iter_i, :]) * 100, np.mean(space_acc[cls_idx, iter_i, :]) * 100))
                for show_idx in show_list:
                    log.info(row.format("average_accuracy",
                                        "[{:>2}, {:>5.2f}]".format(rot_thresh_list[show_idx], trans_thresh_list[show_idx]),
                                        rot_acc[cls_idx, iter_i, show_idx] * 100, trans_acc[cls_idx, iter_i, show_idx] * 100,
                                        space_acc[cls_idx, iter_i, show_idx] * 100))
        for iter_i in range(num_iter):                               # overall performance
            log.info("---------- performance over {} classes -----------".format(num_valid_class))
            log.info("** iter {} **".format(iter_i + 1))
            log.info(head)
            log.info("{:<16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}".format(
                "average_accuracy", "[{:>2}, {:>5.2f}]".format(-1, -1),
                np.sum(rot_acc[:, iter_i, :]) / (num_valid_class * num_metric) * 100,
                np.sum(trans_acc[:, iter_i, :]) / (num_valid_class * num_metric) * 100,
                np.sum(space_acc[:, iter_i, :]) / (num_valid_class * num_metric) * 100))
            for show_idx in show_list:
                log.info(row.format("average_accuracy",
                                    "[{:>2}, {:>5.2f}]".format(rot_thresh_list[show_idx], trans_thresh_list[show_idx]),
                                    np.sum(rot_acc[:, iter_i, show_idx]) / num_valid_class * 100,
                                    np.sum(trans_acc[:, iter_i, show_idx]) / num_valid_class * 100,
                                    np.sum(space_acc[:, iter_i, show_idx]) / num_valid_class * 100))
        return {"rot_acc": rot_acc, "trans_acc": trans_acc, "space_acc": space_acc, "num_valid_class": num_valid_class,
                "rot_thresh": rot_thresh_list, "trans_thresh": trans_thresh_list}

    # ---------------------------------------------------- shared body of the ADD(-S) and arp-2D tables
    def _threshold_tables(self, config, metrics, column, fixed, curve_x, dx, per_class_scale, area_label, area_div,
                          curve_scale, overall_title, blank_after_overall, output_dir, pkl_name):
        log = self.logger
        num_iter = config.TEST.test_iter
        keys = list(fixed)
        count_all = np.zeros((self.num_classes,), dtype=np.float32)
        count_correct = {k: np.zeros((self.num_classes, num_iter), dtype=np.float32) for k in keys}
        thr_mean = np.tile(curve_x.astype(np.float32), (self.num_classes, num_iter, 1))
        count_correct["mean"] = np.zeros((self.num_classes, num_iter, thr_mean.shape[-1]), dtype=np.float32)
        thr_fixed = {k: np.zeros((self.num_classes, num_iter), dtype=np.float32) for k in keys}
        for i, cls_name in enumerate(self.classes):
            s = per_class_scale(cls_name)
            for k in keys:
                thr_fixed[k][i, :] = fixed[k] * s
            thr_mean[i, :, :] *= s
        num_valid_class = 0
        for cls_idx, cls_name in enumerate(self.classes):
            if cls_name not in metrics:
                continue
            num_valid_class += 1
            err = metrics[cls_name][:, :, column]                               # (num_iter, n) float64
            count_all[cls_idx] = err.shape[1]
            for iter_i in range(num_iter):
                e = err[iter_i][:, None]
                for k in keys:
                    count_correct[k][cls_idx, iter_i] = np.sum(e[:, 0] < thr_fixed[k][cls_idx, iter_i])
                count_correct["mean"][cls_idx, iter_i] = np.sum(e < thr_mean[cls_idx, iter_i][None, :], axis=0)
        plot_data, sums = {}, {k: np.zeros(num_iter) for k in ["mean"] + keys}
        acc = {k: np.zeros((self.num_classes, num_iter)) for k in ["mean"] + keys}
        for cls_idx, cls_name in enumerate(self.classes):
            if count_all[cls_idx] == 0:
                continue
            plot_data[cls_name] = []
            for iter_i in range(num_iter):
                log.info("** {}, iter {} **".format(cls_name, iter_i + 1))
                y = count_correct["mean"][cls_idx, iter_i] / float(count_all[cls_idx])
                acc["mean"][cls_idx, iter_i] = _simpson(y, dx) / area_div * 100
                sums["mean"][iter_i] += acc["mean"][cls_idx, iter_i]
                for k in keys:
                    acc[k][cls_idx, iter_i] = 100 * float(count_correct[k][cls_idx, iter_i]) / float(count_all[cls_idx])
                    sums[k][iter_i] += acc[k][cls_idx, iter_i]
                plot_data[cls_name].append((curve_x.astype(np.float32), y if curve_scale == 1.0 else
                                            curve_scale * count_correct["mean"][cls_idx, iter_i] / float(count_all[cls_idx])))
                log.info("{}, area: {:.2f}".format(area_label, acc["mean"][cls_idx, iter_i]))
                for k in keys:
                    log.info("threshold={}, correct poses: {}, all poses: {}, accuracy: {:.2f}".format(
                        k, count_correct[k][cls_idx, iter_i], count_all[cls_idx], acc[k][cls_idx, iter_i]))
                log.info(" ")
        if output_dir is not None:
            with open(os.path.join(output_dir, pkl_name), "wb") as f:
                pickle.dump(plot_data, f, protocol=2)
        log.info("=" * 30)
        for iter_i in range(num_iter):
            log.info(overall_title.format(num_valid_class))
            log.info("** iter {} **".format(iter_i + 1))
            log.info("{}, area: {:.2f}".format(area_label, sums["mean"][iter_i] / num_valid_class))
            for k in keys:
                log.info("threshold={}, mean accuracy: {:.2f}".format(k, sums[k][iter_i] / num_valid_class))
            if blank_after_overall:
                log.info(" ")
        log.info("=" * 30)
        return {"count_all": count_all, "count_correct": count_correct, "accuracy": acc, "curves": plot_data,
                "num_valid_class": num_valid_class}

    # ------------------------------------------------------------------------------- ADD / ADD-S, :372-512
    def evaluate_pose_add(self, config, all_poses_est, all_poses_gt, output_dir=None, metrics=None):
        self.logger.info("evaluating pose add")
        metrics = metrics if metrics is not None else self.pose_metrics(config, all_poses_est, all_poses_gt)
        eval_method = "adi" if any(c in SYM_ADI for c in self.classes if c in metrics) else "add"     # :381, :426 (sticky)
        return self._threshold_tables(
            config, metrics, 2, {"0.02": 0.02, "0.05": 0.05, "0.10": 0.10}, np.arange(0, 0.1, 0.0001), 0.0001,
            lambda c: self._diameters[c], "threshold=[0.0, 0.10]", 0.1, 1.0,
            "---------- add performance over {} classes -----------", False, output_dir, "{}_xys.pkl".format(eval_method))

    # ------------------------------------------------------------------------------- arp 2D, :514-674
    def evaluate_pose_arp_2d(self, config, all_poses_est, all_poses_gt, output_dir=None, metrics=None):
        self.logger.info("evaluating pose average re-projection 2d error")
        metrics = metrics if metrics is not None else self.pose_metrics(config, all_poses_est, all_poses_gt)
        return self._threshold_tables(
            config, metrics, 3, {"2": 2.0, "5": 5.0, "10": 10.0, "20": 20.0}, np.arange(0, 50, 0.1), 0.1, lambda c: 1.0,
            "threshold=[0, 50]", 50.0, 100.0,
            "---------- arp 2d performance over {} classes -----------", True, output_dir, "arp_2d_xys.pkl")
