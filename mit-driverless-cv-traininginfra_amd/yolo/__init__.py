"""Drop-in for the reference's CVC-YOLOv3 hot-path modules (models.py, utils/utils.py, utils/parse_config.py)."""
