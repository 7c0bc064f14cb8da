"""Task constants of the reference fruit-fly tasks (`tasks/constants.py:16-37`), same names."""

# Walking constants.
_WALK_CONTROL_TIMESTEP = 2e-3  # s
_WALK_PHYSICS_TIMESTEP = 2e-4
_TERMINAL_LINVEL = 50  # cm/s
_TERMINAL_ANGVEL = 200  # rad/s

# Flight constants.
_FLY_CONTROL_TIMESTEP = 2e-4
_FLY_PHYSICS_TIMESTEP = 5e-5
_BODY_PITCH_ANGLE = 47.5  # deg
_TERMINAL_HEIGHT = 0.2  # cm

_TERMINAL_QACC = 1e14  # mixed units

_WING_PARAMS = {
    "base_freq": 218.0,
    "gainprm": [18, 18, 18],
    "damping": 0.007769230,
    "stiffness": 0.01,
    "fluidcoef": [1.0, 0.5, 1.5, 1.7, 1.0],
    "rel_freq_range": 0.05,
    "num_freqs": 201,
}

# CoM offset from the root joint in thorax coordinates (`tasks/task_utils.py:188,210`).
_ROOT2COM_OFFSET = (-0.03697732, 0.00029205, -0.0142447)
