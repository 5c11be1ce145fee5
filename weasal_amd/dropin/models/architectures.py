from weasal_amd.architectures import KPFCNN, p2p_fitting_regularizer  # noqa: F401
