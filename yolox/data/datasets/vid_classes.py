"""Class-name tables (reference: yolox/data/datasets/vid_classes.py): the 30 ImageNet-VID and 25 OVIS categories."""
VID_classes = ("airplane", "antelope", "bear", "bicycle", "bird", "bus", "car", "cattle", "dog", "domestic_cat", "elephant", "fox",
               "giant_panda", "hamster", "horse", "lion", "lizard", "monkey", "motorcycle", "rabbit", "red_panda", "sheep", "snake",
               "squirrel", "tiger", "train", "turtle", "watercraft", "whale", "zebra")
OVIS_classes = ("Person", "Bird", "Cat", "Dog", "Horse", "Sheep", "Cow", "Elephant", "Bear", "Zebra", "Giraffe", "Poultry",
                "Giant_panda", "Lizard", "Parrot", "Monkey", "Rabbit", "Tiger", "Fish", "Turtle", "Bicycle", "Motorcycle",
                "Airplane", "Boat", "Vehical")
