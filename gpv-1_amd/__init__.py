# Package body of gpv1_amd (see ../gpv1_amd/__init__.py for the import alias).
