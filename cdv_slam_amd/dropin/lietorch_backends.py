"""Drop-in for the reference's `lietorch_backends` extension (cdvslam/lietorch/src/lietorch.cpp:286-316).
Forward ops of SO3 (group id 1) and SE3 (3); backward ops belong to the training path."""
from cdv_slam_amd import ops


def expm(group_id, a):
    return ops.lie_op(group_id, "exp", a)


def logm(group_id, X):
    return ops.lie_op(group_id, "log", X)


def inv(group_id, X):
    return ops.lie_op(group_id, "inv", X)


def mul(group_id, X, Y):
    return ops.lie_op(group_id, "mul", X, Y)


def adj(group_id, X, a):
    return ops.lie_op(group_id, "adj", X, a)


def adjT(group_id, X, a):
    return ops.lie_op(group_id, "adjT", X, a)


def act(group_id, X, p):
    return ops.lie_op(group_id, "act", X, p)


def act4(group_id, X, p):
    return ops.lie_op(group_id, "act4", X, p)


def as_matrix(group_id, X):
    return ops.lie_op(group_id, "matrix", X)


def _training_only(name):
    def f(*args, **kwargs):
        raise NotImplementedError("lietorch_backends.%s is the training path (out of scope)" % name)
    f.__name__ = name
    return f


expm_backward = _training_only("expm_backward")
logm_backward = _training_only("logm_backward")
inv_backward = _training_only("inv_backward")
mul_backward = _training_only("mul_backward")
adj_backward = _training_only("adj_backward")
adjT_backward = _training_only("adjT_backward")
act_backward = _training_only("act_backward")
act4_backward = _training_only("act4_backward")
projector = _training_only("projector")
Jinv = _training_only("Jinv")
