"""Class-name tables (reference: yolox/data/datasets/vid_classes.py): the 30 ImageNet-VID and 25 OVIS categories."""
VID_classes = ("airplane", "antelope", "bear", "bicycle", "bird", "bus", "car", "cattle", "dog", "domestic_cat", "elephant", "fox",
               "giant_panda", "hamster", "horse", "lion", "lizard", "monkey", "motorcycle", "rabbit", "red_panda", "sheep", "snake",
               "squirrel", "tiger", "train", "turtle", "watercraft", "whale", "zebra")
OVIS_classes = ("Person", "Bird", "Cat", "Dog", "Horse", "Sheep", "Cow", "Elephant", "Bear", "Zebra", "Giraffe", "Poultry",
                "Giant_panda", "Lizard", "Parrot", "Monkey", "Rabbit", "Tiger", "Fish", "Turtle", "Bicycle", "Motorcycle",
                "Airplane", "Boat", "Vehical")

# WordNet ids of the 30 ImageNet-VID categories, in the order of VID_classes (the <name> field of the dataset's XML annotations)
VID_wnids = ("n02691156", "n02419796", "n02131653", "n02834778", "n01503061", "n02924116", "n02958343", "n02402425", "n02084071", "n02121808",
             "n02503517", "n02118333", "n02510455", "n02342885", "n02374451", "n02129165", "n01674464", "n02484322", "n03790512", "n02324045",
             "n02509815", "n02411705", "n01726692", "n02355227", "n02129604", "n04468005", "n01662784", "n04530566", "n02062744", "n02391049")
