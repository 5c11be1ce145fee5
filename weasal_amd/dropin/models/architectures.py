from weasal_amd.architectures import KPFCNN, KPFCNN_mprm, p2p_fitting_regularizer  # noqa: F401
