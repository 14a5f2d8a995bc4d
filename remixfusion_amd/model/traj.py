"""The four anchor fields of the reference's ``Trajectory`` that moving_volume mutates
(model/traj.py:28-31); trajectory file writers are out of scope."""


class Trajectory:
    def __init__(self, dir_path=None):
        self.path = dir_path
        self.pose_list = []
        self.kfx = 0.0
        self.kfy = 0.0
        self.kfz = 0.0
        self.first = 0
