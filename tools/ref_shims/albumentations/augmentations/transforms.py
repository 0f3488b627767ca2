class GridDistortion:
    pass
