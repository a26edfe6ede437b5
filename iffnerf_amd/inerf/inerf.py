"""inerf/inerf.py of the reference: the se(3) camera offset (:64-92) and the photometric loss (:105).

``find_POI`` (:38-49, SIFT key points through OpenCV) feeds only the "interest_points" / "interest_regions" pixel
samplers; OpenCV is not part of this image and ``pose_estimation/test.py:208`` asks for "random", so it raises."""
import torch


def find_POI(img_rgb):
    raise RuntimeError("find_POI needs OpenCV's SIFT (inerf/inerf.py:38-49); only sampling_strategy='random' "
                       "(the one pose_estimation/test.py:208 uses) is available")


def vec2ss_matrix(vector):
    """[w]x, the skew-symmetric matrix of a 3-vector (:52-61)."""
    x, y, z = vector[0], vector[1], vector[2]
    zero = torch.zeros((), dtype=vector.dtype, device=vector.device)
    return torch.stack((torch.stack((zero, -z, y)), torch.stack((z, zero, -x)), torch.stack((-y, x, zero))))


class CameraTransfer(torch.nn.Module):
    """T = exp(theta [w, v]) @ start_pose with the screw-motion exponential of :73-92 (w is NOT renormalised there, so
    neither here).  Parameters start at N(0, 1e-6) like the reference's."""

    def __init__(self, start_pose: torch.Tensor):
        super().__init__()
        self.start_pose = start_pose
        self.w = torch.nn.Parameter(torch.normal(0.0, 1e-6, size=(3,)))
        self.v = torch.nn.Parameter(torch.normal(0.0, 1e-6, size=(3,)))
        self.theta = torch.nn.Parameter(torch.normal(0.0, 1e-6, size=()))

    def forward(self):
        dt, dev = self.start_pose.dtype, self.start_pose.device
        K = vec2ss_matrix(self.w).to(dt)
        K2 = K @ K
        eye = torch.eye(3, dtype=dt, device=dev)
        s, c = torch.sin(self.theta), torch.cos(self.theta)
        rot = eye + s * K + (1 - c) * K2
        trans = (eye * self.theta + (1 - c) * K + (self.theta - s) * K2) @ self.v.to(dt)
        top = torch.cat((rot, trans[:, None]), dim=1)
        bottom = torch.cat((torch.zeros(1, 3, dtype=dt, device=dev), torch.ones(1, 1, dtype=dt, device=dev)), dim=1)   # no host copy: capturable
        return torch.cat((top, bottom), dim=0) @ self.start_pose


def img2mse(x, y):
    return torch.mean((x - y) ** 2)
