"""Drop-in ``models`` package: same import paths, class names, forward signatures and state_dict keys as the reference's
``models/`` tree for the RGI hot path, with the arithmetic running in libe4s_hip.so."""
