"""CPU: host-side rules of emd_amd.graphs (no launches)."""
import types

import pytest
import torch


def _cam(cam_no=0, time_diff=0.0):
    return types.SimpleNamespace(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, world_view_transform=torch.eye(4),
                                 full_proj_transform=torch.eye(4), camera_center=torch.zeros(3), cam_no=cam_no, time_diff=time_diff)


def test_step_inputs_refuses_views_that_differ_in_cam_no_or_time_diff():
    """model.render reads `cam_no` (the row of S3Gaussian's per-camera time_offset, scene/deformation.py:439-451) and `time_diff` from the
    camera object as HOST constants; a one-graph-for-all-views step would bake row 0's values into every view."""
    from emd_amd.graphs import StepInputs
    with pytest.raises(ValueError, match="cam_no"):
        StepInputs([_cam(0), _cam(1)], torch.zeros(3), device="cpu")
    with pytest.raises(ValueError, match="time_diff"):
        StepInputs([_cam(2, 0.1), _cam(2, 0.2)], torch.zeros(3), device="cpu")
    si = StepInputs([_cam(2, 0.1), _cam(2, 0.1)], torch.zeros(3), device="cpu")          # tables only: no launch
    assert si.camera.cam_no == 2 and abs(si.camera.time_diff - 0.1) < 1e-12 and si.rows == 2
